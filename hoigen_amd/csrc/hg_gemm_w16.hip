// Persistent fp16 MFMA GEMM, 256 x 256 x 64 tiles, SIXTEEN waves per workgroup (four per SIMD), gfx950.
//
//   D[m][n] = sum_k A[m][k] * W[n][k]      A: activations [M,K] fp16, W: nn.Linear weight [N,K] fp16
//
// Experiment in latency hiding by occupancy instead of by schedule: the ring kernels (hg_gemm_ring*.hip) run two waves
// per SIMD in a hand-placed [fetch | MFMA] alternation whose fetch segments are longer than its MFMA segments; the duo
// kernel (hg_gemm_duo.hip) runs two independent workgroups per CU.  Here one 1024-thread workgroup holds the
// 256 x 256 accumulator tile in 16 waves of 64 x 64 (64 accumulator VGPRs, <= 128 VGPRs per wave), so that every SIMD
// has four waves to pick from: while one wave is blocked issuing a global->LDS piece or waiting for its fragments,
// others issue MFMAs.  The tile keeps the 256 x 256 operand intensity (64 DMA pieces per 2048 MFMA cycles).
//
// Waves 4(M) x 4(N); LDS ring of two stages (A 32 KiB + W 32 KiB each, XOR-swizzled through the DMA source address).
// Per K-tile and wave:  counted wait for its 4 pieces of tile g (the 4 pieces of tile g+1 stay in flight) -> barrier B1 ->
// fragments of k-step 0 (8 ds_read_b128) -> 16 MFMAs -> fragments of k-step 1 -> barrier B2 (stage read by everyone)
// -> 4 DMA pieces of tile g+2 into the stage -> 16 MFMAs.  Persistent over tiles like the other kernels.
#include <stdio.h>
#include <stdlib.h>

#include "hg_gemm_dev.h"

namespace hg {

template <int EPI>
__global__ __launch_bounds__(1024, 4) void gemm_w16(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                    const unsigned a_bytes, const int gsz) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 256, BK = 32;                         // K-step of 32 = one MFMA k-step; four stages of 32 KiB
    constexpr int HALF = 16384, STAGE = 32768, NST = 4;
    constexpr bool F16OUT = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [stage][A 32 KiB | W 32 KiB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    const int G = gridDim.x, bid = blockIdx.x;
    const bool xcd_ok = (G & 7) == 0;
    const int cpx = xcd_ok ? (G >> 3) : G;
    const int T8 = xcd_ok ? (n_tiles + 7) / 8 : n_tiles;
    const int xbase = xcd_ok ? (bid & 7) * T8 : 0;
    const int xend = (xbase + T8 < n_tiles) ? xbase + T8 : n_tiles;
    const int slot = xbase + (xcd_ok ? (bid >> 3) : bid);
    const int my_tiles = slot < xend ? (xend - slot + cpx - 1) / cpx : 0;
    const int tiles_m_all = n_tiles / tiles_n;
    const int ngf = tiles_n / gsz, grem = tiles_n - ngf * gsz, per_grp = tiles_m_all * gsz;
    auto tile_of = [&](int item, int& tm, int& tn) {
        if (item < ngf * per_grp) {
            const int grp = item / per_grp, rr = item - grp * per_grp;
            tm = rr / gsz;
            tn = grp * gsz + (rr - tm * gsz);
        } else {
            const int rr = item - ngf * per_grp;
            tm = rr / grem;
            tn = ngf * gsz + (rr - tm * grem);
        }
    };
    if (my_tiles <= 0) return;
    const int S = my_tiles * nk;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    // DMA: a piece is 16 rows x 64 B; wave w moves piece w (rows 16w .. 16w+15) of the A tile and of the W tile.
    // lane -> (row l>>2, LDS chunk l&3), source chunk (l&3) ^ ((row>>2)&3)
    const int prow = lane >> 2;
    const int pc = (lane & 3) ^ ((prow >> 2) & 3);
    const int voA0 = prow * p.lda * 2 + pc * 16, voW0 = prow * p.K * 2 + pc * 16;
    const int rowA16 = 16 * p.lda * 2, rowW16 = 16 * p.K * 2;
    int ld_r = 0, ld_kt = 0, ld_oA, ld_oW;
    {
        int tm, tn;
        tile_of(slot, tm, tn);
        ld_oA = tm * BM * p.lda * 2;
        ld_oW = tn * 256 * p.K * 2;
    }
    auto ld_advance = [&]() {
        if (++ld_kt == nk) {
            ld_kt = 0;
            ++ld_r;
            if (ld_r < my_tiles) {
                int tm, tn;
                tile_of(slot + ld_r * cpx, tm, tn);
                ld_oA = tm * BM * p.lda * 2;
                ld_oW = tn * 256 * p.K * 2;
            }
        }
    };
    auto issue = [&](int stage) {
        HG_LDS void* dA = (HG_LDS void*)(smem + stage * STAGE + wave * 1024);
        HG_LDS void* dW = (HG_LDS void*)(smem + stage * STAGE + HALF + wave * 1024);
        const int sA = ld_oA + ld_kt * (BK * 2) + wave * rowA16;
        const int sW = ld_oW + ld_kt * (BK * 2) + wave * rowW16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dA, 16, voA0, sA, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dW, 16, voW0, sW, 0, 0);
    };

    // fragment: lane (row r = l&15, k = 8q .. 8q+7) -> chunk q of the 64-byte row, stored at chunk q ^ ((row>>2)&3)
    const int coff = (((lane >> 4) ^ ((lane >> 2) & 3)) << 4);
    const int a_row = (wm * 64 + (lane & 15)) * 64 + coff;          // + f * 1024
    const int w_row = HALF + (wn * 64 + (lane & 15)) * 64 + coff;   // + g * 1024

    half8 xa[4], wb[4];
    f32x4 acc[4][4];

    for (int i = 0; i < 3 && i < S; ++i) {
        issue(i);
        ld_advance();
    }

    const int q = lane >> 4;
    int g = 0;
    for (int r = 0; r < my_tiles; ++r) {
        int tm, tn;
        tile_of(slot + r * cpx, tm, tn);
        const int m0 = tm * BM, n0 = tn * 256;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) acc[f][gg] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt, ++g) {
            const char* st = smem + (g & (NST - 1)) * STAGE;
            // own pieces of step g landed; steps g+1, g+2 (2 pieces each) may stay in flight
            if (g + 2 < S) wait_vm<4>();
            else if (g + 1 < S) wait_vm<2>();
            else wait_vm<0>();
            barrier_raw();       // every wave's pieces landed - and every wave is done with stage (g-1) & 3
            __builtin_amdgcn_sched_barrier(0);
            if (g + 3 < S) {
                issue((g + 3) & (NST - 1));
                ld_advance();
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) xa[f] = *reinterpret_cast<const half8*>(st + a_row + f * 1024);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) wb[gg] = *reinterpret_cast<const half8*>(st + w_row + gg * 1024);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    acc[f][gg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[gg], xa[f], acc[f][gg], 0, 0, 0);
        }

        // ---------------- epilogue: rows m0 + wm*64 + f*16 + (lane&15), columns n0 + wn*64 + gg*16 + 4q .. +3
        __builtin_amdgcn_sched_barrier(0);
        const int nw = n0 + wn * 64 + 4 * q;
        const bool interior = m0 + BM <= p.M;
        f32x4 bvv[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg)
            bvv[gg] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + gg * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (F16OUT) {
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 vx = acc[2 * pr][gg] + bvv[gg], vy = acc[2 * pr + 1][gg] + bvv[gg];
                    if constexpr (EPI == EPI_BIAS_QGELU_F16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vx[e] = quick_gelu_r(vx[e]); vy[e] = quick_gelu_r(vy[e]); }
                    }
                    if constexpr (EPI == EPI_BIAS_RELU_F16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { vx[e] = fmaxf(vx[e], 0.f); vy[e] = fmaxf(vy[e], 0.f); }
                    }
                    half4 hx, hy;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hx[e] = (half_t)vx[e]; hy[e] = (half_t)vy[e]; }
                    const u32x2 ux = __builtin_bit_cast(u32x2, hx), uy = __builtin_bit_cast(u32x2, hy);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                    const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                    const int m = m0 + wm * 64 + pr * 32 + (lane & 15) + ((q & 1) ? 16 : 0);
                    if (interior || m < p.M)
                        *reinterpret_cast<u32x4*>(outp + (size_t)m * p.ldc + n0 + wn * 64 + 4 * (q & ~1) + gg * 16) = o;
                }
        } else {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    epilogue_ring<EPI>(p, m0 + wm * 64 + f * 16 + (lane & 15), nw + gg * 16, acc[f][gg] + bvv[gg]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

template <int EPI>
static hipError_t launch_w16_t(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 4 * 32768;
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w16<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int n_cu = n_cu_d[dev_i];
    const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 256;
    const int n_tiles = tiles_m * tiles_n;
    const int grid = n_tiles < n_cu ? n_tiles : n_cu;
    const size_t a_bytes = (size_t)tiles_m * 256 * a.lda * 2;
    int gsz = (int)((1536 * 1024) / ((size_t)512 * a.K));
    if (gsz < 3) gsz = 3;
    if (gsz > tiles_n) gsz = tiles_n;
    const int ngroups = (tiles_n + gsz - 1) / gsz;
    gsz = (tiles_n + ngroups - 1) / ngroups;
    hipLaunchKernelGGL((gemm_w16<EPI>), dim3(grid), dim3(1024), LDS, s, a, tiles_n, n_tiles, (unsigned)a_bytes, gsz);
    return hipGetLastError();
}

bool gemm_w16_ok(int epi, const GemmArgs& a) {
    if (!gemm_ring_ok(a)) return false;
    return epi == EPI_BIAS_F16 || epi == EPI_BIAS_QGELU_F16 || epi == EPI_BIAS_RELU_F16 || epi == EPI_BIAS_F32 ||
           epi == EPI_BIAS_RELU_F32;
}

hipError_t launch_gemm_w16(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS_F16: return launch_w16_t<EPI_BIAS_F16>(a, s);
        case EPI_BIAS_QGELU_F16: return launch_w16_t<EPI_BIAS_QGELU_F16>(a, s);
        case EPI_BIAS_RELU_F16: return launch_w16_t<EPI_BIAS_RELU_F16>(a, s);
        case EPI_BIAS_F32: return launch_w16_t<EPI_BIAS_F32>(a, s);
        case EPI_BIAS_RELU_F32: return launch_w16_t<EPI_BIAS_RELU_F32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hg
