// Persistent fp16 MFMA GEMM, 128 x 256 x 64 tiles, TWO independent 4-wave workgroups per CU (gfx950).
//
//   D[m][n] = sum_k A[m][k] * W[n][k]      A: activations [M,K] fp16, W: nn.Linear weight [N,K] fp16
//
// Why a second structure next to hg_gemm_ring*.hip: the ring kernels run ONE 8-wave workgroup per CU whose two wave
// groups alternate [fetch | MFMA] segments in lockstep through workgroup barriers.  That schedule is bound by its
// fetch segments (a wave is blocked while the CU's address unit takes its global->LDS pieces, ~29 cycles per 1 KiB,
// ~35 B/clk/CU), and every CU reaches its epilogue at the same time with the matrix pipe idle - for the residual
// GEMMs (out_proj, c_proj: fp32 read-modify-write + fp16 copy + row statistics, 330 KB per tile) that is an HBM
// burst of the whole chip followed by an HBM-idle K loop.  Here each CU hosts two workgroups of 4 waves (one wave
// per SIMD each, <= 256 VGPRs, 80 KiB of LDS each) that share nothing and never synchronise with each other:
// while one workgroup issues DMA, waits at its barrier or runs its epilogue, the other one's wave on the same SIMD
// feeds the matrix pipe; tile boundaries of the two drift apart by themselves, so epilogue traffic of one overlaps
// the K loop of the other and HBM sees a steady stream instead of bursts.
//
// Workgroup: 256 threads = 4 waves as 2(M) x 2(N); a wave owns 64 rows x 128 columns = acc[4][8] tiles of 16x16
// (128 accumulator VGPRs).  LDS (exactly 80 KiB): one A slot (128 rows x 128 B) + two W slots (256 rows x 128 B).
// Per K-tile:   counted wait for this wave's pieces of A(t), W(t) (W(t+1) stays in flight) -> barrier B1 ->
//               ALL fragments of the K-tile into registers (8 A + 16 W ds_read_b128 per wave) -> barrier B2 (both
//               slots are free again) -> issue A(t+1) into the A slot and W(t+2) into the W slot just read ->
//               64 MFMAs from registers.  The registers are the third pipeline stage: A has a full K-tile of lead,
//               W two.
// The workgroup is persistent and treats the K-tiles of all its tiles as one stream: the first K-tile of the next
// tile is in flight during an epilogue.  Same conventions as the ring kernels: W rows are the MFMA A operand and
// activation rows the B operand (a lane holds 4 consecutive output columns of one row), XOR-swizzled LDS images
// written by buffer_load ... lds through the SOURCE address, n-group-major tile order per XCD.
//
// Epilogues walk the wave's eight 16-column groups: bias (one 16-byte load per group), then the four row tiles.
// Same arithmetic, in the same order, as the ring and simple kernels -> bit-identical outputs; the LayerNorm
// statistics of EPI_RESID_LN_F32 are emitted per contiguous 64-column group (the ring2 kernel's groups are two
// 32-column halves 128 apart: same (mean, rstd) after finalize_stats up to fp32 rounding).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

// Diagnostic build (HG_EXTRA_FLAGS=-DHG_STAMPS): per-wave s_memtime totals per K-tile of 0 top vmcnt wait, 1 barrier B1,
// 2 fragment reads (issue + lgkmcnt), 3 barrier B2, 4 DMA issue, 5 MFMAs, 7 epilogue (per tile); written to GemmArgs::dbg
#ifdef HG_STAMPS
#define DSEG_B() do { __builtin_amdgcn_sched_barrier(0); t_beg = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DSEG_E(k) do { __builtin_amdgcn_sched_barrier(0); tacc[k] += __builtin_amdgcn_s_memtime() - t_beg; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DSEG_B() do {} while (0)
#define DSEG_E(k) do {} while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_duo(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                   const unsigned a_bytes, const int gsz, const int xm) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BK = 64;
    constexpr int A_BYTES = 16384, W_BYTES = 32768;
    constexpr bool SCALED = (EPI == EPI_SCALE_RESID_LN_F32);       // update multiplied by pos[n] (adapter up_proj)
    constexpr bool RLN = (EPI == EPI_RESID_LN_F32 || SCALED);
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_F32 || RLN);
    constexpr bool F16OUT = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [A 16 KiB][W0 32 KiB][W1 32 KiB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = p.K / BK;

    // ---- tile list of this workgroup (as hg_gemm_ring2.hip: n-group-major, XCD x owns [x*T8, (x+1)*T8))
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
    const int S = my_tiles * nk;                                   // K-tiles in this workgroup's stream
#ifdef HG_EXPERIMENTS
    const int xmode = xm;        // timing experiments (wrong results): 1 every tile streams A tile 0, 2 epilogue on tile 0's rows
#else
    constexpr int xmode = 0;
#endif

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    // ---- DMA: a piece is 8 rows x 128 B (1 KiB of LDS, lane-linear); lane -> (row l>>3, 16-byte chunk l&7 of the LDS
    // row), its source chunk is (l&7) ^ ((row>>1)&7) with row = 8*piece + (l>>3): only the parity of the piece enters
    int voA[2], voW[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int c = (lane & 7) ^ ((par * 4 + (lane >> 4)) & 7);
        voA[par] = (lane >> 3) * p.lda * 2 + c * 16;
        voW[par] = (lane >> 3) * p.K * 2 + c * 16;
    }
    const int rowA8 = 8 * p.lda * 2, rowW8 = 8 * p.K * 2;         // bytes between consecutive pieces in global memory
    // two load streams (A runs one K-tile ahead of the consumer, W two): next K-tile to fetch and its tile origin
    struct Ld { int r, kt, org; };
    Ld lA{0, 0, 0}, lW{0, 0, 0};
    {
        int tm, tn;
        tile_of(slot, tm, tn);
        lA.org = (xmode & 1) ? 0 : tm * BM * p.lda * 2;
        lW.org = tn * 256 * p.K * 2;
    }
    auto ld_advance = [&](Ld& l, bool isA) {
        if (++l.kt == nk) {
            l.kt = 0;
            ++l.r;
            if (l.r < my_tiles) {
                int tm, tn;
                tile_of(slot + l.r * cpx, tm, tn);
                l.org = isA ? ((xmode & 1) ? 0 : tm * BM * p.lda * 2) : tn * 256 * p.K * 2;
            }
        }
    };
    // all pieces that share an M0 (LDS base) differ by the instruction's immediate, which the hardware also adds to the
    // global address: the scalar offset takes it back out
    auto issue_A = [&]() {
        HG_LDS void* dst = (HG_LDS void*)(smem + wave * 4096);
        const int so = lA.org + lA.kt * (BK * 2) + wave * 4 * rowA8;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voA[0], so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voA[1], so + rowA8 - 1024, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voA[0], so + 2 * rowA8 - 2048, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voA[1], so + 3 * rowA8 - 3072, 3072, 0);
    };
    auto issue_W = [&](int buf) {
        const int so = lW.org + lW.kt * (BK * 2) + wave * 8 * rowW8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            HG_LDS void* dst = (HG_LDS void*)(smem + A_BYTES + buf * W_BYTES + wave * 8192 + h * 4096);
            const int s2 = so + h * 4 * rowW8;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voW[0], s2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voW[1], s2 + rowW8 - 1024, 1024, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voW[0], s2 + 2 * rowW8 - 2048, 2048, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, dst, 16, voW[1], s2 + 3 * rowW8 - 3072, 3072, 0);
        }
    };

    // ---- fragment read offsets (row bases are multiples of 16 -> the swizzle is a lane constant)
    const int sw = (lane >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
    const int a_row = (wm * 64 + (lane & 15)) * 128;               // + f * 2048
    const int w_row = A_BYTES + (wn * 128 + (lane & 15)) * 128;    // + buf * W_BYTES + g * 2048

    half8 xa[4][2], wb[8][2];
    f32x4 acc[4][8];
#ifdef HG_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_beg = 0;
    const unsigned long long t_all = __builtin_amdgcn_s_memtime();
#endif

    // ---- prologue: W(0), A(0), W(1) in this order: the counted waits leave the youngest W tile in flight
    issue_W(0);
    ld_advance(lW, false);
    issue_A();
    ld_advance(lA, true);
    if (S > 1) {
        issue_W(1);
        ld_advance(lW, false);
    }

    const int q = lane >> 4;
    int g = 0;
    for (int r = 0; r < my_tiles; ++r) {
        int tm, tn;
        tile_of(slot + r * cpx, tm, tn);
        const int m0 = (xmode & 2) ? 0 : tm * BM, n0 = tn * 256;
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) acc[f][gg] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kt = 0; kt < nk; ++kt, ++g) {
            const int wbuf = A_BYTES + (g & 1) * W_BYTES;
            // This wave's pieces of A(g) and W(g) have landed; W(g+1) (the 8 youngest DMAs, when it exists) stays in
            // flight.  Right after a residual epilogue nothing has to be waited for: its loads are younger than those
            // DMAs and their data has been used.  After a store-only epilogue the stores are the youngest operations:
            // the same count is stricter than needed, never too lax.
            DSEG_B();
            if (!(RESID && kt == 0 && r > 0)) {
                if (g + 1 < S) wait_vm<8>();
                else wait_vm<0>();
            }
            DSEG_E(0);
            DSEG_B();
            barrier_raw();                                          // B1: every wave's pieces landed
            DSEG_E(1);
            __builtin_amdgcn_sched_barrier(0);
            DSEG_B();
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xa[f][ks] = *reinterpret_cast<const half8*>(smem + a_row + f * 2048 + coff[ks]);
#pragma unroll
            for (int gg = 0; gg < 8; ++gg)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wb[gg][ks] = *reinterpret_cast<const half8*>(smem + wbuf + (w_row - A_BYTES) + gg * 2048 + coff[ks]);
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): this wave holds its fragments
            DSEG_E(2);
            DSEG_B();
            barrier_raw();                                          // B2: every wave does -> A slot and W slot g&1 are free
            DSEG_E(3);
            __builtin_amdgcn_sched_barrier(0);
            DSEG_B();
            if (g + 1 < S) {
                issue_A();
                ld_advance(lA, true);
            }
            if (g + 2 < S) {
                issue_W(g & 1);
                ld_advance(lW, false);
            }
            DSEG_E(4);
            __builtin_amdgcn_sched_barrier(0);
            DSEG_B();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int gg = 0; gg < 8; ++gg)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[f][gg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[gg][ks], xa[f][ks], acc[f][gg], 0, 0, 0);
            DSEG_E(5);
        }

        DSEG_B();
        // ---------------- epilogue: rows m0 + wm*64 + f*16 + (lane&15), columns n0 + wn*128 + gg*16 + 4q .. +3
        __builtin_amdgcn_sched_barrier(0);
        const int nw = n0 + wn * 128 + 4 * q;
        int mrow[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) mrow[f] = m0 + wm * 64 + f * 16 + (lane & 15);
        const bool interior = m0 + BM <= p.M;
        if constexpr (RESID) {
            float* xo = reinterpret_cast<float*>(p.out);
            size_t ro[4];
            float muv[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const int mc = mrow[f] < p.M ? mrow[f] : p.M - 1;
                ro[f] = (size_t)mc * p.ldc + nw;
                if constexpr (RLN) muv[f] = p.mu[mc];
            }
            // residual rows (+ bias) through a register pipeline DEPTH column groups deep: the fragment registers are
            // dead here, and a shallower pipeline exposes one HBM round trip per group (8 per tile: measured 53 k cycles
            // per tile against 13 k for the whole K loop of out_proj)
            // 5 spills with the LayerNorm extras (row sums, centres), 4 with the column scale of the adapter epilogue on top
            constexpr int DEPTH = RLN ? (SCALED ? 3 : 4) : 5;
            f32x4 xr[8][4], bvv[8], scv[SCALED ? 8 : 1];
            const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
            auto fetch = [&](auto GG) {
                constexpr int gg = decltype(GG)::value;
#pragma unroll
                for (int f = 0; f < 4; ++f) xr[gg][f] = *reinterpret_cast<const f32x4*>(xo + ro[f] + gg * 16);
                bvv[gg] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + gg * 16) : z4;
                if constexpr (SCALED) scv[gg] = *reinterpret_cast<const f32x4*>(p.pos + nw + gg * 16);
            };
            float sum[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            auto consume = [&](auto GG) {
                constexpr int gg = decltype(GG)::value;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    f32x4 u = acc[f][gg] + bvv[gg];
                    if constexpr (SCALED) u *= scv[gg];
                    const f32x4 v = xr[gg][f] + u;
                    acc[f][gg] = v;
                    if (interior || mrow[f] < p.M) *reinterpret_cast<f32x4*>(xo + ro[f] + gg * 16) = v;
                    if constexpr (RLN) sum[f][gg >> 2] += (v[0] + v[1]) + (v[2] + v[3]);
                }
            };
            using std::integral_constant;
            auto step = [&](auto GG) {                      // consume group gg, refill its registers with group gg + DEPTH
                constexpr int gg = decltype(GG)::value;
                consume(GG);
                if constexpr (gg + DEPTH < 8) fetch(integral_constant<int, gg + DEPTH>{});
                __builtin_amdgcn_sched_barrier(0);
            };
            fetch(integral_constant<int, 0>{}); fetch(integral_constant<int, 1>{}); fetch(integral_constant<int, 2>{});
            if constexpr (DEPTH > 3) fetch(integral_constant<int, 3>{});
            if constexpr (DEPTH > 4) fetch(integral_constant<int, 4>{});
            __builtin_amdgcn_sched_barrier(0);
            step(integral_constant<int, 0>{}); step(integral_constant<int, 1>{}); step(integral_constant<int, 2>{});
            step(integral_constant<int, 3>{}); step(integral_constant<int, 4>{}); step(integral_constant<int, 5>{});
            step(integral_constant<int, 6>{}); step(integral_constant<int, 7>{});
            if constexpr (RLN) {
                // per row and 64-column group: (sum, sum of squared deviations from the group mean) for finalize_stats
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const int sg = tn * 4 + wn * 2;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float s = sum_rows(sum[f][h]);
                        const float gm = s * (1.0f / 64.0f);
                        float m2 = 0.f;
#pragma unroll
                        for (int gg = 4 * h; gg < 4 * h + 4; ++gg)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float d = acc[f][gg][e] - gm;
                                m2 = fmaf(d, d, m2);
                            }
                        m2 = sum_rows(m2);
                        if (q == 0 && (interior || mrow[f] < p.M))
                            *reinterpret_cast<f32x2*>(p.stats + ((size_t)mrow[f] * p.stats_ld + sg + h) * 2) = f32x2{s, m2};
                    }
                // centred fp16 copy: row tiles (2pr, 2pr+1) paired through v_permlane16_swap -> 16-byte stores
                half_t* out2 = p.out2;
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int m = m0 + wm * 64 + pr * 32 + (lane & 15) + ((q & 1) ? 16 : 0);
                    const size_t o2 = (size_t)m * p.ldc + n0 + wn * 128 + 4 * (q & ~1);
#pragma unroll
                    for (int gg = 0; gg < 8; ++gg) {
                        half4 hx, hy;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hx[e] = (half_t)(acc[2 * pr][gg][e] - muv[2 * pr]);
                            hy[e] = (half_t)(acc[2 * pr + 1][gg][e] - muv[2 * pr + 1]);
                        }
                        const u32x2 ux = __builtin_bit_cast(u32x2, hx), uy = __builtin_bit_cast(u32x2, hy);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        if (interior || m < p.M) *reinterpret_cast<u32x4*>(out2 + o2 + gg * 16) = o;
                    }
                }
            }
        } else if constexpr (F16OUT) {
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            f32x4 bvv[8];                                  // all eight bias vectors in flight at once (one round trip)
#pragma unroll
            for (int gg = 0; gg < 8; ++gg)
                bvv[gg] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + gg * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
                const f32x4 bv = bvv[gg];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    f32x4 vx = acc[2 * pr][gg] + bv, vy = acc[2 * pr + 1][gg] + bv;
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
                        *reinterpret_cast<u32x4*>(outp + (size_t)m * p.ldc + n0 + wn * 128 + 4 * (q & ~1) + gg * 16) = o;
                }
            }
        } else {
            f32x4 bvv[8];
#pragma unroll
            for (int gg = 0; gg < 8; ++gg)
                bvv[gg] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nw + gg * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gg = 0; gg < 8; ++gg) {
#pragma unroll
                for (int f = 0; f < 4; ++f) epilogue_ring<EPI>(p, mrow[f], nw + gg * 16, acc[f][gg] + bvv[gg]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        DSEG_E(7);
    }
#ifdef HG_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 4 + wave) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = tacc[k];
        d[8] = __builtin_amdgcn_s_memtime() - t_all;
        d[9] = (unsigned long long)S;
        d[10] = (unsigned long long)my_tiles;
    }
#endif
#endif
}

template <int EPI>
static hipError_t launch_duo_t(const GemmArgs& a, hipStream_t s) {
    constexpr int LDS = 16384 + 2 * 32768;                          // exactly 80 KiB: two workgroups per CU
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_duo<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int n_cu = n_cu_d[dev_i];
    const int tiles_m = (a.M + 127) / 128, tiles_n = a.N / 256;
    const int n_tiles = tiles_m * tiles_n;
    const size_t a_bytes = (size_t)tiles_m * 128 * a.lda * 2;
#ifdef HG_EXPERIMENTS
    static const int gsz_env = []() { const char* e = getenv("HG_RING_GSZ"); return e ? atoi(e) : 0; }();
#else
    constexpr int gsz_env = 0;
#endif
    int gsz = gsz_env > 0 ? gsz_env : (int)((1536 * 1024) / ((size_t)512 * a.K));
    if (gsz < 3) gsz = 3;
    if (gsz > tiles_n) gsz = tiles_n;
    if (gsz_env <= 0) {
        const int ngroups = (tiles_n + gsz - 1) / gsz;
        gsz = (tiles_n + ngroups - 1) / ngroups;
    }
    // timing experiments: HG_DUO_LDS_CUT = bytes requested less (results wrong), HG_DUO_GRID = workgroups per CU
#ifdef HG_EXPERIMENTS
    static const int xm_env = []() { const char* e = getenv("HG_DUO_XMODE"); return e ? atoi(e) : 0; }();
    static const int lds_cut = []() { const char* e = getenv("HG_DUO_LDS_CUT"); return e ? atoi(e) : 0; }();
    static const int per_cu = []() { const char* e = getenv("HG_DUO_GRID"); const int v = e ? atoi(e) : 2; return v >= 1 ? v : 2; }();
#else
    constexpr int xm_env = 0, lds_cut = 0, per_cu = 2;
#endif
    const int grid2 = n_tiles < per_cu * n_cu ? n_tiles : per_cu * n_cu;
#ifdef HG_STAMPS
    if (getenv("HG_STAMPS")) {
        const size_t n = (size_t)grid2 * 4 * 16;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipMemsetAsync(d, 0, n * 8, s);
        GemmArgs b = a;
        b.dbg = d;
        hipLaunchKernelGGL((gemm_duo<EPI>), dim3(grid2), dim3(256), LDS - lds_cut, s, b, tiles_n, n_tiles, (unsigned)a_bytes, gsz, xm_env);
        hipStreamSynchronize(s);
        unsigned long long* h = (unsigned long long*)malloc(n * 8);
        hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
        static const char* names[8] = {"vmcnt", "B1", "reads", "B2", "DMA-issue", "MFMA", "-", "epilogue/tile"};
        double acc[11] = {0};
        for (size_t w = 0; w < (size_t)grid2 * 4; ++w)
            for (int k = 0; k < 11; ++k) acc[k] += (double)h[w * 16 + k];
        const double kts = acc[9] > 0 ? acc[9] : 1, tl = acc[10] > 0 ? acc[10] : 1;
        fprintf(stderr, "[stamps] duo<%d> N=%d K=%d: kernel %.0f cycles per wave, %.0f per K-tile;", EPI, a.N, a.K,
                acc[8] / (grid2 * 4.0), acc[8] / kts);
        for (int k = 0; k < 8; ++k)
            if (names[k][0] != '-') fprintf(stderr, " %s %.0f", names[k], acc[k] / (k == 7 ? tl : kts));
        fprintf(stderr, "\n");
        free(h);
        hipFree(d);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((gemm_duo<EPI>), dim3(grid2), dim3(256), LDS - lds_cut, s, a, tiles_n, n_tiles, (unsigned)a_bytes, gsz, xm_env);
    return hipGetLastError();
}

bool gemm_duo_ok(int epi, const GemmArgs& a) {
    // as the ring kernels, but any K >= 64 (the K loop has no per-position specialisation)
    if (a.N % 256 || a.K % 64 || a.K < 64 || a.M < 512 || a.N > 8192) return false;
    const size_t Mp = (size_t)((a.M + 255) / 256) * 256;
    if (Mp * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;
    if (epi == EPI_RESID_LN_F32 || epi == EPI_SCALE_RESID_LN_F32)
        return a.out2 && a.stats && a.mu && a.stats_ld == 4 * (a.N / 256) && (epi == EPI_RESID_LN_F32 || a.pos);
    return epi == EPI_BIAS_F16 || epi == EPI_BIAS_QGELU_F16 || epi == EPI_BIAS_RELU_F16 || epi == EPI_BIAS_RESID_F32 ||
           epi == EPI_BIAS_F32 || epi == EPI_BIAS_RELU_F32 || epi == EPI_PATCH_F32;
}

hipError_t launch_gemm_duo(int epi, const GemmArgs& a, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS_F16: return launch_duo_t<EPI_BIAS_F16>(a, s);
        case EPI_BIAS_QGELU_F16: return launch_duo_t<EPI_BIAS_QGELU_F16>(a, s);
        case EPI_BIAS_RELU_F16: return launch_duo_t<EPI_BIAS_RELU_F16>(a, s);
        case EPI_BIAS_RESID_F32: return launch_duo_t<EPI_BIAS_RESID_F32>(a, s);
        case EPI_BIAS_F32: return launch_duo_t<EPI_BIAS_F32>(a, s);
        case EPI_BIAS_RELU_F32: return launch_duo_t<EPI_BIAS_RELU_F32>(a, s);
        case EPI_PATCH_F32: return launch_duo_t<EPI_PATCH_F32>(a, s);
        case EPI_RESID_LN_F32: return launch_duo_t<EPI_RESID_LN_F32>(a, s);
        case EPI_SCALE_RESID_LN_F32: return launch_duo_t<EPI_SCALE_RESID_LN_F32>(a, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hg
