// Device-side helpers shared by the gfx950 kernels (wave64, MFMA fp16, LDS swizzles).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define HG_MAX_DEVICES 16
// index of the calling thread's current HIP device for per-device launcher state (clamped)
inline int current_device_index() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= HG_MAX_DEVICES) d = 0;
    return d;
}

#define HG_LDS __attribute__((address_space(3)))
#define HG_GLOBAL __attribute__((address_space(1)))

// 16-byte asynchronous global -> LDS copy (LDS destination = wave-uniform base + lane*16).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const HG_GLOBAL void*)gsrc, (HG_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): give every XCD a contiguous range of
// logical tile ids so neighbouring tiles share one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
