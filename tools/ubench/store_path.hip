// Micro-benchmark: bytes per clock one CU can STORE to global memory with global_store_dwordx4 (16 B per lane), by
// address pattern and by the number of CUs storing at the same time.  8 waves per CU, each wave streams `iters` store
// instructions into its own 64 KiB window (L2-resident, rewritten over and over) or into a private 16 MiB region (HBM).
//   patterns: 0 lane-linear (1 KiB contiguous per instruction = 8 full lines)
//             1 GEMM fp16 epilogue: 32 rows x 32 B, row stride 4608 B (32 partial lines per instruction)
//             2 fp32 epilogue: 16 rows x 64 B, row stride 3072 B
//   hipcc --offload-arch=gfx950 -O3 -o store_path store_path.hip && ./store_path
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(char* __restrict__ dst, size_t region, size_t stride, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* base = dst + (size_t)blockIdx.x * stride;
    size_t lane_off;
    size_t step;      // bytes the wave advances per instruction
    if (PATTERN == 0) { lane_off = (size_t)lane * 16; step = 1024; }
    else if (PATTERN == 1) { lane_off = (size_t)(lane & 31) * 4608 + (lane >> 5) * 16; step = 32; }
    else { lane_off = (size_t)(lane & 15) * 3072 + (lane >> 4) * 16; step = 64; }
    const size_t wave_region = region / 8;
    char* wbase = base + (size_t)wave * wave_region;
    const u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    size_t off = 0;
    const size_t span = PATTERN == 0 ? 1024 : (PATTERN == 1 ? (size_t)32 * 4608 : (size_t)16 * 3072);
    for (int it = 0; it < iters; ++it) {
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(wbase + off + lane_off), "v"(v) : "memory");
        off += step;
        if (PATTERN == 0) { if (off + span > wave_region) off = 0; }
        else if (PATTERN == 1) { if ((off % 4608) == 0 || off + span > wave_region) off = 0; }   // stay inside the row
        else { if ((off % 3072) == 0 || off + span > wave_region) off = 0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int PATTERN> static void run(char* dst, size_t region, size_t stride, int grid, int iters, const char* what, int khz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    store_kernel<PATTERN><<<grid, 512>>>(dst, region, stride, iters / 8);
    (void)hipEventRecord(e0);
    store_kernel<PATTERN><<<grid, 512>>>(dst, region, stride, iters);
    (void)hipEventRecord(e1);
    hipError_t rc = hipEventSynchronize(e1);
    if (rc != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(rc)); exit(1); }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 8 * iters * 1024.0;
    printf("%-34s grid %3d  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU  %6.1f clk per store instruction per CU\n", what, grid, ms,
           bytes / ms * 1e-9, bytes / (ms * 1e-3) / (khz * 1e3) / grid, (ms * 1e-3) * (khz * 1e3) / (8.0 * iters));
    fflush(stdout);
}

int main() {
    int cus = 0, khz = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    char* dst;
    const size_t total = (size_t)4 << 30;
    (void)hipMalloc(&dst, total);
    (void)hipMemset(dst, 0, total);
    const int iters = 4096;
    for (int grid : {cus, 32, 8}) {
        // L2-resident windows: 8 waves x 160 KiB (patterns 1 / 2 need 32 x 4608 B per wave)
        const size_t reg_l2 = (size_t)8 * 160 * 1024, reg_hbm = (size_t)16 << 20;
        run<0>(dst, reg_l2, reg_l2, grid, iters, "linear 1 KiB, L2 window", khz);
        run<1>(dst, reg_l2, reg_l2, grid, iters, "32 rows x 32 B, L2 window", khz);
        run<2>(dst, reg_l2, reg_l2, grid, iters, "16 rows x 64 B, L2 window", khz);
        run<0>(dst, reg_hbm, reg_hbm, grid, iters, "linear 1 KiB, 16 MiB per CU (HBM)", khz);
        run<1>(dst, reg_hbm, reg_hbm, grid, iters, "32 rows x 32 B, 16 MiB per CU", khz);
    }
    (void)hipFree(dst);
    return 0;
}
