// Residual GEMMs of the ViT block on a LOCK-STEP, software-pipelined K loop (gfx950):
//
//   x[m][n] += sum_k A[m][k] W[n][k] + bias[n]   (+ fp16 copy and row statistics: EPI_RESID_LN_F32; plain: EPI_BIAS_RESID_F32)
//
// Geometry of hg_gemm_ring2.hip (128 x 256 x 64 tiles, 8 waves as 2 x 4, three 48 KiB LDS stages filled by
// buffer_load ... lds, persistent workgroups walking their tiles as one K-tile stream, the same epilogues) - but a different
// schedule.  ring2 alternates [fetch | 16 MFMAs] segments between the two waves of a SIMD, four barriers per K-tile; its
// in-kernel stamps (round 3) show a wave spending 1180 of 2750 cycles per K-tile issuing its fetch segments and 750 at
// barriers against 540 in MFMAs: every segment exposes an LDS round trip, a DMA-issue queue and a barrier.  Here all eight
// waves run the same STEP (one k-step of 32: 16 MFMAs per wave) at the same time, two steps and two barriers per K-tile:
//
//   step (t, k0):  16 MFMAs on fragments (t, k0)  ||  read fragments (t, k1)      ||  DMA: second half of K-tile t+2
//   step (t, k1):  16 MFMAs on fragments (t, k1)  ||  read fragments (t+1, k0)    ||  DMA: first half of K-tile t+3
//   every step ends with lgkmcnt(0), (even steps: the counted vmcnt that says K-tile t+1 has landed) and one s_barrier.
//
// The LDS reads of the NEXT step and the DMA issue sit between the MFMAs of the current one, both waves of a SIMD feed the
// matrix pipe concurrently, and nothing but the barrier itself is exposed.  Stage (t+2) % 3 is the stage of K-tile t-1, whose
// last fragments were read during step (t-1, k0): it is free from step (t-1, k1) on.  A K-tile's DMA pieces are issued
// 2-3 steps before the barrier that publishes them.
// Same accumulation order per accumulator as the ring / simple kernels: bit-identical results.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_flow(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                    const unsigned a_bytes, const int mode, const int gsz) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BK = 64;
    constexpr int AB = 16384, WH = 16384;              // bytes: A tile (both halves), one W half
    constexpr int STAGE = AB + 2 * WH;                 // 48 KiB
    constexpr int NST = 3;
    constexpr int H1 = 3, H2 = 3;                      // DMA instructions per wave: first half (A0 A1 W0), second half (W1 W2 W3)
    constexpr int NY = H1 + H2;                        // younger DMAs when K-tile t+1 must have landed (end of step (t, k0))
    constexpr bool RLN = (EPI == EPI_RESID_LN_F32);
    static_assert(RLN || EPI == EPI_BIAS_RESID_F32, "residual epilogues only");
    constexpr int E = RLN ? 28 : 16;                   // epilogue store instructions per wave
    constexpr int RA = RLN ? 10 : 8, RB = RLN ? 10 : 8;   // residual-row (+ row centre) prefetch loads per wave, two steps
    constexpr int BIAS_OFF = NST * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    // ---- tile list (as in ring2): XCD-contiguous chunks, column tile of a row panel fastest
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
    {      // start stagger of the workgroups with slack (see ring2)
        const int dunit = mode >> 8;
        const int max_tiles = (T8 + cpx - 1) / cpx;
        if (dunit > 0 && my_tiles < max_tiles) {
            const unsigned h = ((unsigned)bid * 2654435761u) >> 24;
            const long long d = ((long long)(max_tiles - my_tiles) * nk * dunit * h) >> 8;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < d) __builtin_amdgcn_s_sleep(32);
        }
    }

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    // ---- DMA source offsets: a piece is 8 rows x 128 B; lane -> (row = l>>3, chunk' = l&7)
    int voffA[2], voffW[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave * 2 + i) * 8 + (lane >> 3);            // 0..127
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffA[i] = row * p.lda * 2 + c * 16 - i * 1024;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);            // 0..255 (W0 = 0..127, W1 = 128..255)
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffW[i] = row * p.K * 2 + c * 16 - i * 1024;
    }
    // ---- load stream: one cursor; a K-tile's pieces are issued in two consecutive steps (first half, second half)
    struct Ld { int kt, r, sA, sW, st; };
    Ld ld{nk - 1, -1, 0, 0, (NST - 1) * STAGE};
    // WRAP: 0 = stays inside its output tile, 1 = moves to the next output tile, 2 = decide at run time (prologue)
    auto advance = [&](auto WRAP_T) {
        constexpr int WRAP = decltype(WRAP_T)::value;
        ++ld.kt;
        if (WRAP == 1 || (WRAP == 2 && ld.kt == nk)) {
            ld.kt = 0;
            ++ld.r;
            int tm, tn;
            tile_of(slot + ld.r * cpx, tm, tn);
            ld.sA = tm * BM * p.lda * 2;
            ld.sW = tn * 256 * p.K * 2;
#ifdef FLOW_TILE0      // timing experiment: every tile streams tile 0 (operands from L2)
            ld.sA = 0; ld.sW = 0;
#endif
#ifdef FLOW_A0         // timing experiment: the A panel of tile 0 only (A from L2, W as it is)
            ld.sA = 0;
#endif
        }
        ld.st = ld.st == (NST - 1) * STAGE ? 0 : ld.st + STAGE;
    };
    auto dma_A = [&](auto I) {
        constexpr int i = decltype(I)::value;
#ifdef FLOW_NO_DMA
        return;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + ld.st + wave * 2 * 1024), 16, voffA[i],
                                                 ld.sA + ld.kt * (BK * 2), i * 1024, 0);
    };
    auto dma_W = [&](auto I) {
        constexpr int i = decltype(I)::value;
#ifdef FLOW_NO_DMA
        return;
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + ld.st + AB + wave * 4 * 1024), 16, voffW[i],
                                                 ld.sW + ld.kt * (BK * 2), i * 1024, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;
    auto issue_first = [&](auto WRAP_T) {      // advances to the next K-tile of the stream
        advance(WRAP_T);
        dma_A(P0{}); dma_A(P1{}); dma_W(P0{});
    };
    auto issue_second = [&]() { dma_W(P1{}); dma_W(P2{}); dma_W(P3{}); };
    using WDYN = std::integral_constant<int, 2>;

    // ---- fragment read offsets (row bases are multiples of 16 -> lane-constant swizzle)
    const int sw = (lane >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
    const int a_row = (wm * 32 + (lane & 15)) * 128;                 // + ha*8192 + f*2048
    const int w_row = AB + (wn * 32 + (lane & 15)) * 128;            // + hb*WH + g2*2048

    half8 fa[2][2][2], fw[2][2][2];                                   // [k-step][ha][f], [k-step][hb][g2]
    auto read_fa = [&](auto KS, int st) {
        constexpr int ks = decltype(KS)::value;
#ifdef FLOW_NO_READ
        return;
#endif
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
            for (int f = 0; f < 2; ++f)
                fa[ks][ha][f] = *reinterpret_cast<const half8*>(smem + st + ha * 8192 + a_row + f * 2048 + coff[ks]);
    };
    auto read_fw = [&](auto KS, int st) {
        constexpr int ks = decltype(KS)::value;
#ifdef FLOW_NO_READ
        return;
#endif
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
                fw[ks][hb][g2] = *reinterpret_cast<const half8*>(smem + st + hb * WH + w_row + g2 * 2048 + coff[ks]);
    };
    auto read_frags = [&](auto KS, int st) { read_fa(KS, st); read_fw(KS, st); };
    f32x4 acc[2][2][2][2];
    auto mma = [&](auto KS, auto HA) {
        constexpr int ks = decltype(KS)::value, ha = decltype(HA)::value;
#ifdef FLOW_NO_MFMA
#pragma unroll
        for (int f = 0; f < 2; ++f) asm volatile("" ::"v"(fa[ks][ha][f]), "v"(fw[ks][f][0]), "v"(fw[ks][f][1]));
        return;
#endif
        {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
                        acc[ha][hb][f][g2] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks][hb][g2], fa[ks][ha][f], acc[ha][hb][f][g2], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // [8 MFMAs with 4 fragment reads of the next step between them]: one read per two MFMAs
    auto interleave = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    };
    auto end_step = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the next step's fragments are in registers, the stage is read
#ifndef FLOW_NO_BARRIER
        barrier_raw();
#endif
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- bias -> LDS once per workgroup
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < p.N / 4; i += 512)
            *reinterpret_cast<f32x4*>(smem + BIAS_OFF + i * 16) = p.bias ? reinterpret_cast<const f32x4*>(p.bias)[i] : z;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // ---- prologue: K-tiles 0 and 1 complete, first half of K-tile 2; fragments (0, k0)
    issue_first(WDYN{}); issue_second();
    issue_first(WDYN{}); issue_second();
    issue_first(WDYN{});
    wait_vm<NY + H1>();                        // K-tile 0 landed
    barrier_raw();
    read_frags(I0{}, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);

    const bool late = wave >= 4;
    int stg = 0;                               // LDS stage of the current K-tile
    bool prev_full = false;
    for (int r = 0; r < my_tiles; ++r) {
        int tm, tn;
        tile_of(slot + r * cpx, tm, tn);
        const int m0 = tm * BM, n0 = tn * 256;
        const bool post_ok = prev_full;        // the previous tile issued all E epilogue stores (it lay inside M)
        prev_full = m0 + BM <= p.M;
        const bool more = r + 1 < my_tiles;    // another output tile follows in this workgroup's stream
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) acc[a][b][f][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xres[2][2][2][2];
        float muv[RLN ? 2 : 1][RLN ? 2 : 1];
        auto load_xres = [&](int ha) {          // residual rows (and row centres) of A half `ha` of this tile
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                int m = m0 + ha * 64 + wm * 32 + f * 16 + (lane & 15);
                m = m < p.M ? m : p.M - 1;
                if constexpr (RLN) muv[ha][f] = p.mu[m];
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * (lane >> 4);
                        xres[ha][hb][f][g2] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.out) + (size_t)m * p.ldc + n);
                    }
            }
        };
        // One K-tile = two steps.  KIND: 0 middle, 1 first of a tile (the previous epilogue's stores may be pending), 2 / 3 / 4
        // the third-to-last, second-to-last (residual prefetch) and last K-tile of a tile - only there the refills and the
        // waits depend on whether another tile follows.
        auto ktile = [&](auto KIND_T) {
            constexpr int KIND = decltype(KIND_T)::value;
            const int st = stg * STAGE;
            stg = stg == NST - 1 ? 0 : stg + 1;
            const int nst = stg * STAGE;       // stage of the next K-tile of the stream
            // ---------------- step (t, k0): second half of K-tile t+2 (the cursor was advanced by the previous step)
            // DMA issue blocks the issuing wave while the CU's address unit works through the pieces: waves 0-3 (one per SIMD)
            // issue in front of their MFMAs, waves 4-7 between their two halves - a SIMD always has a wave feeding the pipe
            if (!late) {
                if (KIND < 3 || more) issue_second();
                if constexpr (KIND == 3) load_xres(0);
            }
            mma(I0{}, I0{});
            read_fa(I1{}, st);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (late) {
                if (KIND < 3 || more) issue_second();
                if constexpr (KIND == 3) load_xres(0);
            }
            mma(I0{}, I1{});
            read_fw(I1{}, st);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            // K-tile t+1 must have landed: all but the DMA instructions issued since its last piece
            if constexpr (KIND == 0 || KIND == 2) wait_vm<NY>();
            else if constexpr (KIND == 1) { if (post_ok) wait_vm<NY + E>(); else wait_vm<NY>(); }
            else if constexpr (KIND == 3) { if (more) wait_vm<NY + RA>(); else wait_vm<RA>(); }
            else { if (more) wait_vm<NY + RA + RB>(); }          // last K-tile: K-tile t+1 is the next tile's first
            end_step();
            // ---------------- step (t, k1): first half of K-tile t+3
            if (!late) {
                if (KIND < 2 || more) issue_first(std::integral_constant<int, KIND == 2 ? 1 : 0>{});
                if constexpr (KIND == 3) load_xres(1);
            }
            mma(I1{}, I0{});
            if (KIND < 4 || more) read_fa(I0{}, nst);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (late) {
                if (KIND < 2 || more) issue_first(std::integral_constant<int, KIND == 2 ? 1 : 0>{});
                if constexpr (KIND == 3) load_xres(1);
            }
            mma(I1{}, I1{});
            if (KIND < 4 || more) read_fw(I0{}, nst);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            end_step();
        };
        {
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
            using K3 = std::integral_constant<int, 3>;
            using K4 = std::integral_constant<int, 4>;
            ktile(K1{});
            for (int kt = 1; kt < nk - 3; ++kt) ktile(K0{});
            ktile(K2{});
            ktile(K3{});
            ktile(K4{});
        }
        // ---------------- epilogue (ring2's): the fragments of the next tile's first step stay in their registers
        const int q = lane >> 4;
        auto epilogue = [&](auto INTERIOR_T) {
        constexpr bool INTERIOR = decltype(INTERIOR_T)::value;
        if constexpr (RLN) {
            half_t* out2 = p.out2;
            const int sg = tn * 4 + wn;                     // column group of this wave
#pragma unroll
            for (int ha = 0; ha < 2; ++ha) {
                half4 h16[2][2][2];                         // [f][hb][g2]
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int m = m0 + ha * 64 + wm * 32 + f * 16 + (lane & 15);
                    f32x4 v[2][2];
                    float sum = 0.f;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * q;
                            v[hb][g2] = xres[ha][hb][f][g2] +
                                        (acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4));
                            sum += (v[hb][g2][0] + v[hb][g2][1]) + (v[hb][g2][2] + v[hb][g2][3]);
                            if (INTERIOR || m < p.M)
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = v[hb][g2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) h16[f][hb][g2][e] = (half_t)(v[hb][g2][e] - muv[ha][f]);
                        }
                    sum = sum_rows(sum);
                    const float gm = sum * (1.0f / 64.0f);
                    float m2 = 0.f;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float d = v[hb][g2][e] - gm;
                                m2 = fmaf(d, d, m2);
                            }
                    m2 = sum_rows(m2);
                    if (q == 0 && (INTERIOR || m < p.M)) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<f32x2*>(p.stats + ((size_t)m * p.stats_ld + sg) * 2) = f32x2{sum, m2};
                    }
                }
                // fp16 copy: pair the row tiles f = 0, 1 through v_permlane16_swap -> 16-byte stores
                const int mX = m0 + ha * 64 + wm * 32 + (lane & 15);
                const int m = mX + ((q & 1) ? 16 : 0);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int nb = n0 + hb * 128 + wn * 32 + g2 * 16;
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x2 ux = __builtin_bit_cast(u32x2, h16[0][hb][g2]), uy = __builtin_bit_cast(u32x2, h16[1][hb][g2]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        if (INTERIOR || m < p.M) *reinterpret_cast<u32x4*>(out2 + (size_t)m * p.ldc + nb + 4 * (q & ~1)) = o;
                    }
            }
        } else {
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int m = m0 + ha * 64 + wm * 32 + f * 16 + (lane & 15);
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * q;
                            const f32x4 v = acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4);
                            if (INTERIOR || m < p.M)
                                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) =
                                    xres[ha][hb][f][g2] + v;
                        }
                }
        }
        };
        if (m0 + BM <= p.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

template <int EPI>
static hipError_t launch_flow_t(const GemmArgs& a, hipStream_t s) {
    constexpr int RING = 3 * 49152;
    const int LDS = RING + a.N * 4;
    if (LDS > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set_d[HG_MAX_DEVICES] = {};      // function attributes and CU counts are per device
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    bool& attr_set = attr_set_d[dev_i];
    int& n_cu = n_cu_d[dev_i];
    if (!attr_set) {
        n_cu = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_flow<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles_m = (a.M + 127) / 128, tiles_n = a.N / 256;
    const int n_tiles = tiles_m * tiles_n;
    const int grid = n_tiles < n_cu ? n_tiles : n_cu;
    const size_t a_bytes = (size_t)tiles_m * 128 * a.lda * 2;
    static const int mode = []() {
        const char* d = getenv("HG_FLOW_DELAY");      // start stagger: estimated cycles per K-tile, 0 = off
        return (d ? atoi(d) : 1200) << 8;
    }();
    static const int gsz_env = []() { const char* e = getenv("HG_RING_GSZ"); return e ? atoi(e) : 0; }();
    int gsz = gsz_env > 0 ? gsz_env : (int)((1536 * 1024) / ((size_t)512 * a.K));
    if (gsz < 3) gsz = 3;
    if (gsz > tiles_n) gsz = tiles_n;
    if (gsz_env <= 0) {
        const int ngroups = (tiles_n + gsz - 1) / gsz;
        gsz = (tiles_n + ngroups - 1) / ngroups;
    }
    hipLaunchKernelGGL((gemm_flow<EPI>), dim3(grid), dim3(512), LDS, s, a, tiles_n, n_tiles, (unsigned)a_bytes, mode, gsz);
    return hipGetLastError();
}

// as ring2 (N % 256 == 0, K % 64 == 0, M >= 512, 32-bit offsets), K >= 320 (five K-tiles: the kinds are distinct K-tiles)
bool gemm_flow_ok(int epi, const GemmArgs& a) {
    if (!gemm_ring2_ok(a) || a.K < 320) return false;
    if (epi == EPI_RESID_LN_F32) return a.out2 && a.stats && a.mu && a.stats_ld == 4 * (a.N / 256);
    return epi == EPI_BIAS_RESID_F32;
}

hipError_t launch_gemm_flow(int epi, const GemmArgs& a, hipStream_t s) {
    if (!gemm_flow_ok(epi, a)) return hipErrorInvalidValue;
    return epi == EPI_RESID_LN_F32 ? launch_flow_t<EPI_RESID_LN_F32>(a, s) : launch_flow_t<EPI_BIAS_RESID_F32>(a, s);
}

}  // namespace hg
