// Probe of DPP row controls / bank masks on gfx950: prints, for each control, which lane every lane reads from.
// hipcc --offload-arch=gfx950 -O2 dpp_probe.hip -o ../../ab/dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int BANK>
__global__ void probe(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xF, BANK, false);
}
template <int CTRL, int BANK>
void run(const char* name) {
    int* d; int h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL((probe<CTRL, BANK>), dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("%-28s:", name);
    for (int i = 0; i < 16; ++i) printf(" %2d", h[i]);
    printf(" | lanes 16..19:");
    for (int i = 16; i < 20; ++i) printf(" %2d", h[i]);
    printf("\n");
    hipFree(d);
}
int main() {
    run<0x128, 0xF>("row_ror:8 bank 1111");
    run<0x128, 0xC>("row_ror:8 bank 1100");
    run<0x128, 0x3>("row_ror:8 bank 0011");
    run<0x114, 0xA>("row_shr:4 bank 1010");
    run<0x104, 0x5>("row_shl:4 bank 0101");
    run<0x124, 0xF>("row_ror:4 bank 1111");
    run<0x12C, 0xF>("row_ror:12 bank 1111");
    return 0;
}
