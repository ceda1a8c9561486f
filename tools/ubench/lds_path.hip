// Micro-benchmark: how many bytes per clock does one CU move from global memory (L2 / MALL / HBM) towards LDS,
// by path?  mode 0: global_load_lds_dwordx4 (the DMA the GEMM kernels use); mode 1: global_load_dwordx4 into
// VGPRs (consumed by an xor); mode 2: global_load_dwordx4 + ds_write_b128 (the register-staged path).
// One 512-thread workgroup per CU (LDS-limited), every wave keeps two batches of P 1-KiB pieces in flight.
//   hipcc --offload-arch=gfx950 -O3 -o lds_path lds_path.hip && ./lds_path
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// region: bytes each workgroup walks (wraps); stride: byte distance between the regions of consecutive workgroups
template <int MODE, int P, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const char* __restrict__ src, size_t region, size_t stride,
                                                             int iters, uint32_t* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const char* base = src + (size_t)blockIdx.x * stride;
    const size_t step = (size_t)WAVES * P * 1024;  // bytes per workgroup iteration
    size_t off = (size_t)wave * P * 1024;
    char* my_lds = lds + (size_t)wave * 2 * P * 1024;
    u32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            char* dst = my_lds + (it & 1) * P * 1024;
#pragma unroll
            for (int p = 0; p < P; ++p)
                __builtin_amdgcn_global_load_lds((const GLB_AS void*)(base + off + p * 1024 + lane * 16),
                                                 (LDS_AS void*)(dst + p * 1024), 16, 0, 0);
            wait_vm<P>();  // the previous batch has landed
            off += step;
            if (off + (size_t)P * 1024 > region) off = (size_t)wave * P * 1024;
        }
        wait_vm<0>();
    } else {
        // inline asm throughout: the compiler would otherwise delete all but the last iteration
        u32x4 r[2][P];
        const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)my_lds + lane * 16;
        auto issue = [&](int h) {
            const char* g = base + off + lane * 16;
#pragma unroll
            for (int p = 0; p < P; ++p)
                asm volatile("global_load_dwordx4 %0, %1, off offset:0" : "=&v"(r[h][p]) : "v"(g + p * 1024) : "memory");
        };
        auto consume = [&](int h) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (MODE == 1)
                    // all four dwords are read after the wait: a dword the compiler sees as dead would be
                    // re-allocated while the load is still in flight
                    asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %2\n v_xor_b32 %0, %0, %3\n v_xor_b32 %0, %0, %4"
                                 : "+v"(acc.x)
                                 : "v"(r[h][p].x), "v"(r[h][p].y), "v"(r[h][p].z), "v"(r[h][p].w));
                else
                    asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (h * P + p) * 1024), "v"(r[h][p]) : "memory");
            }
            if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        issue(0);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                off += step;
                if (off + (size_t)P * 1024 > region) off = (size_t)wave * P * 1024;
                issue(h ^ 1);
                wait_vm<P>();
                consume(h);
            }
        }
        wait_vm<0>();
        consume(0);
    }
    __syncthreads();
    if (MODE != 1) acc = *(const u32x4*)(lds + threadIdx.x * 16);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = acc.x;  // keeps the loads alive
}

template <int MODE, int P, int WAVES>
static float run(const char* src, size_t region, size_t stride, int iters, uint32_t* sink, int grid, int lds_bytes) {
    auto k = stream_kernel<MODE, P, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<<<grid, WAVES * 64, lds_bytes>>>(src, region, stride, iters / 8, sink);
    hipEventRecord(e0);
    k<<<grid, WAVES * 64, lds_bytes>>>(src, region, stride, iters, sink);
    hipEventRecord(e1);
    hipError_t rc = hipEventSynchronize(e1);
    if (rc != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(rc)); fflush(stdout); exit(1); }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return ms;
}

int main(int argc, char** argv) {
    int dev = 0, cus = 0, khz = 0;
    hipSetDevice(dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    const int grid = (argc > 1) ? atoi(argv[1]) : cus;
    const size_t total = (size_t)4 << 30;
    char* src;
    uint32_t* sink;
    hipMalloc(&src, total);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, total);
    hipDeviceSynchronize();
    struct Case {
        const char* name;
        size_t region, stride;
    };
    // shared 1 MiB: every workgroup walks the same L2-resident megabyte (the W operand);
    // private 64 KiB: L2-resident per workgroup; private 16 MiB: streams from HBM (the A operand)
    const Case cases[] = {{"shared-1MiB (L2)", (size_t)1 << 20, 0},
                          {"private-64KiB (L2)", (size_t)64 << 10, (size_t)64 << 10},
                          {"private-1MiB (MALL)", (size_t)1 << 20, (size_t)1 << 20},
                          {"private-16MiB (HBM)", (size_t)16 << 20, (size_t)16 << 20}};
    printf("device: %d CUs, %d MHz; grid %d workgroups\n", cus, khz / 1000, grid);
    printf("%-22s %-28s %10s %10s %12s\n", "case", "path", "ms", "TB/s", "B/clk/CU");
    fflush(stdout);
    const int iters = 2048;
    for (const Case& c : cases) {
#define RUN(MODE, P, WAVES, label)                                                                            \
    {                                                                                                          \
        float ms = run<MODE, P, WAVES>(src, c.region, c.stride, iters, sink, grid, WAVES * 2 * P * 1024);      \
        double bytes = (double)grid * iters * WAVES * P * 1024.0;                                              \
        printf("%-22s %-28s %10.3f %10.2f %12.1f\n", c.name, label, ms, bytes / ms * 1e-9,                     \
               bytes / (ms * 1e-3) / ((double)khz * 1e3) / grid);                                              \
        fflush(stdout);                                                                                        \
    }
        RUN(0, 8, 8, "dma->lds 8w x 8 pieces");
        RUN(0, 4, 8, "dma->lds 8w x 4 pieces");
        RUN(0, 8, 4, "dma->lds 4w x 8 pieces");
        RUN(0, 4, 16, "dma->lds 16w x 4 pieces");
        RUN(1, 8, 8, "load->vgpr 8w x 8");
        RUN(1, 4, 8, "load->vgpr 8w x 4");
        RUN(1, 8, 4, "load->vgpr 4w x 8");
        RUN(1, 4, 16, "load->vgpr 16w x 4");
        RUN(2, 8, 8, "load->vgpr->ds_write 8w x 8");
        RUN(2, 4, 8, "load->vgpr->ds_write 8w x 4");
        RUN(2, 4, 16, "load->vgpr->ds_write 16w x 4");
    }
    hipFree(src);
    hipFree(sink);
    return 0;
}
