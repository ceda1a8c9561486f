// Micro-benchmark: where do the watts go?  One persistent 512-thread workgroup per CU runs, for a few seconds each, the
// per-K-tile instruction mix of the 256x256x64 ring GEMM with parts of it switched off:
//   bit 0  64 x v_mfma_f32_16x16x32_f16 per wave (operands: the fragments read below, or fixed random registers)
//   bit 1  24 x ds_read_b128 per wave (the A / W fragments of a 128 x 64 wave tile; random fp16 data in LDS)
//   bit 2  8 x buffer_load ... lds per wave (64 KiB per workgroup per K-tile from an L2-resident 1 MiB region)
//   bit 3  the same DMA volume from a private 16 MiB region per workgroup (HBM / MALL)
// Prints achieved TFLOP/s and bytes/clk; tools/gpu_power_mix.sh samples rocm-smi (power, sclk) beside each variant.
//   hipcc --offload-arch=gfx950 -O3 -o power_mix power_mix.hip && ./power_mix <mask> <seconds>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HG_LDS __attribute__((address_space(3)))

template <int MASK>
__global__ __launch_bounds__(512, 2) void mix(const char* __restrict__ src, size_t region, size_t stride, int ktiles, float* out) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill 128 KiB of LDS with pseudo-random fp16 in [-1, 1)
    for (int i = tid; i < 128 * 1024 / 2; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15;
        reinterpret_cast<_Float16*>(lds)[i] = (_Float16)(((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f));
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * stride), 0, (unsigned)region, 0x00020000);
    f32x4 acc[2][2][4][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 2; ++g) acc[a][b][f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    half8 xa[4][2], wb[2][2][2];
    // fixed operands for the variants without fragment reads
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xa[f][ks] = *reinterpret_cast<const half8*>(lds + ((f * 2 + ks) * 64 + lane) * 16);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wb[h][g][ks] = *reinterpret_cast<const half8*>(lds + 8192 + (((h * 2 + g) * 2 + ks) * 64 + lane) * 16);
    unsigned off = 0;
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = (kt & 1) * 65536;
#pragma unroll
        for (int ha = 0; ha < 2; ++ha) {
            if (MASK & 2) {      // this half's A fragments (8 reads) + (first half only) both W halves (8 reads): 24 per K-tile
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        xa[f][ks] = *reinterpret_cast<const half8*>(lds + buf + ha * 16384 + ((wave >> 2) * 64 + f * 16 + (lane & 15)) * 128 + (((ks * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4));
                if (ha == 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                wb[h][g][ks] = *reinterpret_cast<const half8*>(lds + buf + 32768 + h * 16384 + ((wave & 3) * 32 + g * 16 + (lane & 15)) * 128 + (((ks * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) << 4));
                }
            }
            if (MASK & 12) {     // 4 DMA pieces of 1 KiB per wave per half: 64 KiB per workgroup per K-tile
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (HG_LDS void*)(lds + (buf ^ 65536) + ha * 32768 + (wave * 4 + i) * 1024), 16,
                                                             lane * 16, (int)(off % (unsigned)(region - 65536)) + (wave * 4 + i) * 1024 + ha * 32768, 0, 0);
                }
            }
            if (MASK & 1) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
#pragma unroll
                            for (int g = 0; g < 2; ++g)
                                acc[ha][hb][f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[hb][g][ks], xa[f][ks], acc[ha][hb][f][g], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("" ::"v"(xa[f][0]), "v"(xa[f][1]));
#pragma unroll
                for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(wb[h][0][0]), "v"(wb[h][0][1]), "v"(wb[h][1][0]), "v"(wb[h][1][1]));
            }
        }
        off += 65536;
        if (MASK & 12) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the previous K-tile's pieces have landed
            __builtin_amdgcn_s_barrier();
        } else if (MASK & 2) {
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int g = 0; g < 2; ++g) s += acc[a][b][f][g][0] + acc[a][b][f][g][3];
    if (s == 12345.678f) out[0] = s;
}

static size_t g_region_kib = 0;      // argv[3]: private region per workgroup in KiB (with bit 3): 512 -> 128 MB in all = Infinity Cache

template <int MASK>
static void run(const char* src, int grid, double seconds, float* out) {
    const bool hbm = MASK & 8;
    const size_t region = hbm ? (g_region_kib ? g_region_kib << 10 : (size_t)16 << 20) : (size_t)1 << 20, stride = hbm ? region : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<MASK>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int ktiles = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    mix<MASK><<<grid, 512, 128 * 1024>>>(src, region, stride, 2000, out);
    (void)hipDeviceSynchronize();
    double total_ms = 0;
    int launches = 0;
    while (total_ms < seconds * 1e3) {
        (void)hipEventRecord(e0);
        mix<MASK><<<grid, 512, 128 * 1024>>>(src, region, stride, ktiles, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms;
        ++launches;
    }
    const double kt = (double)launches * ktiles * grid;
    const double flops = (MASK & 1) ? kt * 2.0 * 256 * 256 * 64 : 0.0;
    printf("mask %2d [%s%s%s%s]: %.2f us per K-tile, %.1f TFLOP/s, LDS reads %.1f GB/s/CU, DMA %.1f GB/s/CU (%s)\n", MASK,
           (MASK & 1) ? "mfma " : "", (MASK & 2) ? "ds_read " : "", (MASK & 4) ? "dma-L2 " : "", (MASK & 8) ? "dma-HBM " : "",
           total_ms * 1e3 / ((double)launches * ktiles), flops / (total_ms * 1e-3) * 1e-12,
           (MASK & 2) ? 196608.0 * launches * ktiles / (total_ms * 1e-3) * 1e-9 : 0.0,
           (MASK & 12) ? 65536.0 * launches * ktiles / (total_ms * 1e-3) * 1e-9 : 0.0,
           hbm ? (g_region_kib ? "private regions of argv[3] KiB" : "private 16 MiB regions") : "shared 1 MiB");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int mask = argc > 1 ? atoi(argv[1]) : 1;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    if (argc > 3) g_region_kib = (size_t)atoi(argv[3]);
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    char* src;
    float* out;
    const size_t total = (size_t)cus * (16 << 20) + (1 << 20);
    (void)hipMalloc(&src, total);
    (void)hipMalloc(&out, 64);
    std::vector<uint16_t> h((1 << 20) / 2);
    srand(7);
    for (auto& v : h) v = (uint16_t)(0x3000 + (rand() & 0x0FFF) + ((rand() & 1) << 15));      // fp16 values of magnitude 0.1 .. 0.25
    for (size_t o = 0; o < total; o += (1 << 20)) (void)hipMemcpy(src + o, h.data(), 1 << 20, hipMemcpyHostToDevice);
    switch (mask) {
        case 1: run<1>(src, cus, seconds, out); break;
        case 2: run<2>(src, cus, seconds, out); break;
        case 3: run<3>(src, cus, seconds, out); break;
        case 4: run<4>(src, cus, seconds, out); break;
        case 6: run<6>(src, cus, seconds, out); break;
        case 7: run<7>(src, cus, seconds, out); break;
        case 8: run<8>(src, cus, seconds, out); break;
        case 11: run<11>(src, cus, seconds, out); break;
        default: printf("mask must be one of 1 2 3 4 6 7 8 11\n"); return 1;
    }
    return 0;
}
