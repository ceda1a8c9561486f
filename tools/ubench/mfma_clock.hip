// Micro-benchmark: cycles per MFMA (s_memtime, shader clock) and the clock the chip holds (cycles / wall time) for
// back-to-back 16x16x32 and 32x32x16 fp16 MFMAs, at one and two waves per SIMD, on zero and on random operands.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip && ./mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES, int KIND>
__global__ __launch_bounds__(WAVES * 64) void loop(const half8* __restrict__ src, int iters, float* out, unsigned long long* cyc) {
    const half8 a = src[threadIdx.x], b = src[256 + threadIdx.x];
    f32x4 acc[16];
    f32x16 big[4];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) big[i][j] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += big[i][j];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int WAVES, int KIND>
static void run(const half8* src, int grid, int iters, float* out, unsigned long long* cyc, const char* label) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    loop<WAVES, KIND><<<grid, WAVES * 64>>>(src, iters / 8, out, cyc);
    (void)hipEventRecord(e0);
    loop<WAVES, KIND><<<grid, WAVES * 64>>>(src, iters, out, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_iter = KIND == 0 ? 64 : 32, n = (double)iters * per_iter;
    const double flops = (double)grid * WAVES * n * (KIND == 0 ? 16384.0 : 32768.0);
    printf("%-46s %8.2f ms %8.1f TFLOP/s  %6.2f cycles/MFMA/wave  clock %.0f MHz\n", label, ms, flops / ms * 1e-9,
           (double)c / n, (double)c / (ms * 1e3));
    fflush(stdout);
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    unsigned long long* cyc;
    half8 *zero, *rnd;
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&cyc, 64);
    (void)hipMalloc(&zero, 2048 * 16);
    (void)hipMalloc(&rnd, 2048 * 16);
    (void)hipMemset(zero, 0, 2048 * 16);
    std::vector<_Float16> h(2048 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    (void)hipMemcpy(rnd, h.data(), 2048 * 16, hipMemcpyHostToDevice);
    const int it = 200000;   // ~10-20 ms per launch
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(zero, cus, it, out, cyc, "16x16x32 1 wave/SIMD  zeros");
        run<4, 0>(rnd, cus, it, out, cyc, "16x16x32 1 wave/SIMD  random");
        run<8, 0>(zero, cus, it, out, cyc, "16x16x32 2 waves/SIMD zeros");
        run<8, 0>(rnd, cus, it, out, cyc, "16x16x32 2 waves/SIMD random");
        run<4, 1>(zero, cus, it, out, cyc, "32x32x16 1 wave/SIMD  zeros");
        run<4, 1>(rnd, cus, it, out, cyc, "32x32x16 1 wave/SIMD  random");
        run<8, 1>(zero, cus, it, out, cyc, "32x32x16 2 waves/SIMD zeros");
        run<8, 1>(rnd, cus, it, out, cyc, "32x32x16 2 waves/SIMD random");
    }
    return 0;
}
