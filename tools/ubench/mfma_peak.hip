// Micro-benchmark: sustained v_mfma_f32_16x16x32_f16 rate with nothing else going on (no memory, no barriers):
// W waves per CU (1, 2 per SIMD), 16 independent accumulators per wave.  Reports TFLOP/s and the implied clock if the
// pipe issues one MFMA per 16 cycles per SIMD (8 passes x ... = 1024 flop/clk/SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop(int iters, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
// 32x32x16 (8 passes): four independent accumulators per wave
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma32_loop(int iters, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.5f); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;
}
// 16x16x32 with one independent VALU / LDS-read instruction between MFMAs (does co-issue fill the gap?)
template <int WAVES, int FILL>
__global__ __launch_bounds__(WAVES * 64) void mfma_fill_loop(int iters, float* out) {
    __shared__ float lds[4096];
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.5f); }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float v = threadIdx.x;
    f32x4 t = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                if (FILL == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
                if (FILL == 2 && (i & 3) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((threadIdx.x & 255) * 16));
                if (FILL == 3) asm volatile("s_nop 0");
            }
        if (FILL == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(t));
    }
    float s = v + t[0];
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <typename K> static void run_k(K kern, int waves, int grid, int iters, float* out, const char* label, int mhz, double mfma_per_iter, double cyc_per_mfma) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    kern<<<grid, waves * 64>>>(iters / 8, out);
    (void)hipEventRecord(e0);
    kern<<<grid, waves * 64>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * waves * iters * mfma_per_iter, flops = n_mfma * 16384 * (cyc_per_mfma / 16);
    const double cycles = (waves >= 4 ? waves / 4 : 1) * (double)iters * mfma_per_iter * cyc_per_mfma;
    printf("%-40s %8.3f ms %9.1f TFLOP/s  implied clock %.0f MHz (nominal %d)\n", label, ms, flops / ms * 1e-9,
           cycles / (ms * 1e-3) * 1e-6, mhz);
    fflush(stdout);
}

template <int WAVES> static void run(int grid, int iters, float* out, const char* label, int mhz) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    mfma_loop<WAVES><<<grid, WAVES * 64>>>(iters / 8, out);
    (void)hipEventRecord(e0);
    mfma_loop<WAVES><<<grid, WAVES * 64>>>(iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * WAVES * iters * 64.0, flops = n_mfma * 16 * 16 * 32 * 2;
    // per SIMD: (WAVES / 4) waves x iters x 64 MFMAs x 16 cycles
    const double cycles = (WAVES >= 4 ? WAVES / 4 : 1) * (double)iters * 64 * 16;
    printf("%-28s %8.3f ms %9.1f TFLOP/s  implied clock %.0f MHz (nominal %d)\n", label, ms, flops / ms * 1e-9,
           cycles / (ms * 1e-3) * 1e-6, mhz);
    fflush(stdout);
}

int main() {
    int cus = 0, khz = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    float* out;
    (void)hipMalloc(&out, 64);
    const int mhz = khz / 1000;
    run_k(mfma32_loop<4>, 4, cus, 40000, out, "32x32x16: 4 waves/CU", mhz, 32, 32);
    run_k(mfma32_loop<8>, 8, cus, 40000, out, "32x32x16: 8 waves/CU", mhz, 32, 32);
    run_k(mfma_fill_loop<4, 0>, 4, cus, 40000, out, "16x16x32: 4 waves/CU plain", mhz, 64, 16);
    run_k(mfma_fill_loop<4, 1>, 4, cus, 40000, out, "16x16x32: 4 waves/CU + v_add each", mhz, 64, 16);
    run_k(mfma_fill_loop<4, 2>, 4, cus, 40000, out, "16x16x32: 4 waves/CU + ds_read_b128 /4", mhz, 64, 16);
    run_k(mfma_fill_loop<4, 3>, 4, cus, 40000, out, "16x16x32: 4 waves/CU + s_nop each", mhz, 64, 16);
    run_k(mfma_fill_loop<8, 1>, 8, cus, 40000, out, "16x16x32: 8 waves/CU + v_add each", mhz, 64, 16);
    run_k(mfma_fill_loop<8, 2>, 8, cus, 40000, out, "16x16x32: 8 waves/CU + ds_read_b128 /4", mhz, 64, 16);
    run_k(mfma_fill_loop<12, 0>, 12, cus, 40000, out, "16x16x32: 12 waves/CU plain", mhz, 64, 16);
    for (int rep = 0; rep < 1; ++rep) {
        run<4>(cus, 20000, out, "4 waves/CU, short (0.5 ms)", khz / 1000);
        run<8>(cus, 20000, out, "8 waves/CU, short", khz / 1000);
        run<4>(cus, 400000, out, "4 waves/CU, long (10 ms)", khz / 1000);
        run<8>(cus, 400000, out, "8 waves/CU, long", khz / 1000);
        run<8>(cus / 2, 100000, out, "8 waves/CU, half the CUs", khz / 1000);
    }
    return 0;
}
