// Micro-benchmark: cycles per ds_write for the address patterns of the fused in_proj + attention kernel's head -> LDS phase
// (hg_qkv_attn_body.inc: write_head) against linear patterns.  WAVES waves of one workgroup each issue N writes of 13 "row
// blocks" x 3 column blocks; s_memtime around the burst including the final s_waitcnt lgkmcnt(0).
//   pattern 0: the kernel's: lane (q = lane >> 4, r16 = lane & 15) -> row r16 (128-byte rows), 8 bytes at chunk (2 sub + (q >> 1)) ^ swz(r16),
//              half (q & 1); row block rb as a 2 KiB immediate offset               (ds_write_b64)
//   pattern 1: the same rows, 16 bytes per lane (paired row blocks)                 (ds_write_b128)
//   pattern 2: linear: lane * 8 bytes, consecutive 512-byte pieces                  (ds_write_b64)
//   pattern 3: linear: lane * 16 bytes                                              (ds_write_b128)
//   pattern 4: the kernel's rows with the chunk rotated by the row's low bits as well: (chunk ^ swz(r16)) -> ((chunk ^ swz) + 2 (r16 & 1)) & 7
//   hipcc --offload-arch=gfx950 -O3 -o lds_write_pattern lds_write_pattern.hip && ./lds_write_pattern
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }

template <int PAT>
__global__ __launch_bounds__(512) void k(int reps, unsigned long long* out, int waves_active) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r16 = lane & 15;
    u32x4 v = {(uint32_t)lane, 2u, 3u, 4u};
    unsigned long long t = 0;
    __syncthreads();
    if (wave < waves_active) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int l12 = (wave & 3) * 3 + c;
                const int mtx = l12 >> 2, sub = l12 & 3;
                const int chunk = 2 * sub + (q >> 1);
                uint32_t a;
                if (PAT == 0) a = mtx * 26624 + (q & 1) * 8 + r16 * 128 + ((chunk ^ swz_k(r16)) << 4);
                else if (PAT == 1) a = mtx * 26624 + (q & 1) * 2048 + r16 * 128 + ((chunk ^ swz_k(r16)) << 4);
                else if (PAT == 2) a = (wave * 3 + c) * 13 * 512 + lane * 8;
                else if (PAT == 3) a = (wave * 3 + c) * 7 * 1024 + lane * 16;
                else a = mtx * 26624 + (q & 1) * 8 + r16 * 128 + ((((chunk ^ swz_k(r16)) + 2 * (r16 & 1)) & 7) << 4);
                a += (uint32_t)(uintptr_t)(LDS_AS char*)lds;
                if (PAT == 0 || PAT == 4) {
#pragma unroll
                    for (int rb = 0; rb < 13; ++rb)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(u32x2{v.x, v.y}), "n"(rb * 2048) : "memory");
                } else if (PAT == 1) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(j * 4096) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(u32x2{v.x, v.y}), "n"(12 * 2048) : "memory");
                } else if (PAT == 2) {
#pragma unroll
                    for (int rb = 0; rb < 13; ++rb)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(u32x2{v.x, v.y}), "n"(rb * 512) : "memory");
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(j * 1024) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        t = __builtin_amdgcn_s_memtime() - t0;
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t;
}

template <int PAT> static void run(const char* name, int waves) {
    unsigned long long* d;
    hipMalloc(&d, 256 * 8 * 8);
    const int reps = 200;
    hipFuncSetAttribute((const void*)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(512), 128 * 1024, 0, reps, d, waves);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    int n = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) { s += (double)h[b * 8 + w]; ++n; }
    const double cyc = s / n / reps;      // cycles per burst of 39 x 512 B (or the same bytes as b128) per wave
    printf("%-44s waves %d: %7.0f cycles per burst of 19 968 B per wave = %5.1f B/clk per CU\n", name, waves, cyc, waves * 19968.0 / cyc);
    hipFree(d);
}

int main() {
    for (int waves : {1, 4, 8}) {
        run<0>("kernel pattern, ds_write_b64", waves);
        run<1>("kernel pattern, paired ds_write_b128", waves);
        run<2>("linear, ds_write_b64", waves);
        run<3>("linear, ds_write_b128", waves);
        run<4>("kernel rows, chunk rotated by row parity", waves);
    }
    return 0;
}
