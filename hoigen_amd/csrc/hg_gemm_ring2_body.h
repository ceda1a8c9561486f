// Body of the 128 x 256 persistent ring GEMM (hg_gemm_ring2.hip holds the kernel's description, the stand-alone kernel and its launchers)
// as a device function over a tile schedule, so that the MLP pair kernel (hg_mlp_pair.hip) runs the same K loop and residual epilogue on
// tiles whose A panel other workgroups of the launch produce.  SCHED: n_items(), tile(r, tm, tn), slack(), CONSUME (+ poll_blocking(r),
// poll_issue(r), poll_finish(r, value)).
#pragma once
#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

// Diagnostic build (-DHG_STAMPS -DHG_STAMP_MASK=bits): per-wave s_memtime totals, as in hg_gemm_ring.hip
// (0 vmcnt waits, 1 lgkmcnt waits, 2 fetch barriers, 3 MFMA segments, 4 MFMA barriers, 7 epilogue)
#ifdef HG_STAMPS
#ifndef HG_STAMP_MASK
#define HG_STAMP_MASK 0xFF
#endif
#define SEG_B(k) do { if constexpr ((HG_STAMP_MASK >> (k)) & 1) t_beg = __builtin_amdgcn_s_memtime(); } while (0)
#define SEG_E(k) do { if constexpr ((HG_STAMP_MASK >> (k)) & 1) tacc[k] += __builtin_amdgcn_s_memtime() - t_beg; } while (0)
#else
#define SEG_B(k) do {} while (0)
#define SEG_E(k) do {} while (0)
#endif

// HL (EPI_RESID_LN_F32 only): how the residual stream is held on the way in / out (GemmArgs::hl): 0 fp32 / fp32, 1 fp32 /
// hi + lo, 2 hi + lo / hi + lo, 3 hi + lo / fp32.  hi = the centred fp16 copy (row-major: the next GEMM's operand), lo = the
// remainder (x - centre) - hi in tile-fragment order: piece (ha, hb, g2) of a wave = this lane's two row tiles f = 0, 1 side by
// side, 64 lanes contiguous - whole lines in, whole lines out.  lo is bf8 (HG_LO8, hg_kernels.h: 2 x 4 B per lane; 6 bytes per
// element through the epilogue) or fp16 (2 x 8 B; 8 bytes) instead of the fp32 stream's 10, and 16 instead of 24 partial-line
// store instructions per wave and tile.
// GS (EPI_RESID_LN_F32 only; GemmArgs::gamma): the copy the next GEMM reads is fp16((x' - mu) * gamma[n]) - the next LayerNorm's weight
// rides in the ACTIVATION copy, so that the consuming GEMM multiplies by the layer's own fp16 weights (as the reference does) and not by
// a re-rounded fp16(W * gamma): the folded text tower's excess error against the reference was that second rounding
// (tests/test_gpu_text_fold_study.py).  Where the stream leaves as hi + lo (HL 1, 2) the unscaled hi stays the stream's half in out2 and
// the scaled copy goes to out3 (8 more stores per wave and tile); where it leaves as fp32 (HL 0, 3) out2 itself is the scaled copy.
// (The unscaled hi in fragment order beside lo, with out2 always the scaled copy, was built too: the same time, six spilled registers.)
template <int EPI, int HL, bool GS, class SCHED>
__device__ __forceinline__ void gemm_ring2_body(const GemmArgs& p, const int tiles_n, const unsigned a_bytes, const int mode,
                                                const SCHED& sc) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins (buffer resources, LDS DMA): host sees just the stub
    // timing-experiment switches (HG_RING_MODE bits 1 locality, 2 no MFMA, 4 no epilogue, 8 no stagger, 64 no operand DMA)
    // exist only in a -DHG_EXPERIMENTS build: run-time branches in the K loop cost several per cent.  (Round 2's store
    // experiments - lane-linear stores, a tile's read-modify-write trickled under the next tile's K loop as junk accesses -
    // are in the history; results in DESIGN.md 4.)
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int BM = 128, BK = 64;
    constexpr int AB = 16384, WH = 16384;              // bytes: A tile (both halves), one W half
    constexpr int STAGE = AB + 2 * WH;                 // 48 KiB
    constexpr int NST = 3;
    constexpr int GA = 2, GW = 4;                      // DMA instructions per wave: A tile, both W halves
    constexpr int NWT = GW + GA + GW;                  // younger DMAs when A,W(t+1) must have landed
    constexpr bool RLN = (EPI == EPI_RESID_LN_F32);    // residual + fp16 copy + LayerNorm statistics for the next GEMM
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_SCALE_RESID_F32 || RLN);
    constexpr bool F16OUT = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16);
    static_assert(HL == 0 || EPI == EPI_RESID_LN_F32, "hi / lo stream: EPI_RESID_LN_F32 only");
    constexpr bool IN_HL = (HL == 2 || HL == 3), OUT_HL = (HL == 1 || HL == 2);
    static_assert(!GS || EPI == EPI_RESID_LN_F32, "gamma-scaled copy: EPI_RESID_LN_F32 only");
    constexpr int E = F16OUT ? 8 : (RLN ? (OUT_HL ? (GS ? 28 : 20) : 28) : 16);    // epilogue store instructions per wave
    constexpr int R = RESID ? (RLN ? (IN_HL ? 24 : 20) : 16) : 0;      // residual (+ row centre) prefetch loads per wave
    // CON (SCHED::CONSUME, the MLP pair kernel's c_proj): the A operand of a tile is written by OTHER workgroups of this launch on the
    // same XCD (c_fc tiles, plain stores that have reached the XCD's L2 when the counter says so).  Tile r may not touch its A panel before that panel's ready counter has reached its target:
    //   * the first tile: wave 0 polls (sc.poll_blocking) in front of the prologue, a barrier holds the other waves back;
    //   * tile r + 1: wave 0 issues ONE relaxed agent-scope load of the counter six K-tiles before the end of tile r (K-tile kind 5;
    //     one more operation in its vmcnt stream: its waits of kinds 5 and 6 allow for it), the counted wait two K-tiles later retires
    //     it, and at the start of the third-to-last K-tile - one K-tile before the A stream wraps to tile r + 1 - wave 0 looks at the
    //     value (sc.poll_finish: a bounded spin if the panel is not complete yet) in front of the phase's barrier;
    //   * A is fetched with sc1 loads: they bypass this CU's vector L1 and are served by the XCD's L2, where the producers' stores are.
    constexpr bool CON = SCHED::CONSUME;
    static_assert(!CON || EPI == EPI_RESID_LN_F32, "hand-off: the LayerNorm-emitting residual epilogue");
    constexpr int BIAS_OFF = NST * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef HG_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_beg = 0, t_all = 0;
#endif

    // (the MLP pair kernel's schedules rebuild the thread id from the wave index and v_mbcnt, behind an opaque move: threadIdx.x itself
    // would have to stay in v0 across the other body, and with v0 / v1 taken every register tuple of this body - accumulators, fragments -
    // starts at 2 (mod 4) instead of 0: the same K loop then runs 6 % slower (profiles/r06_mlp_pair.txt))
    const int tid = sc.thread_id(), lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    // ---- this workgroup's tiles come from the schedule `sc`: RingTileList (hg_gemm_dev.h) for the stand-alone kernel; the MLP pair
    // kernel (hg_mlp_pair.hip) deals its own list and makes every tile wait for the c_fc tiles of its row panel (SCHED::CONSUME)
    const int bid = blockIdx.x;
    const int my_tiles = sc.n_items();
    auto tile_of = [&](int r, int& tm, int& tn) { sc.tile(r, tm, tn); };
    // De-synchronised epilogues: all tiles take the same time, so every CU would store (and, for the residual
    // epilogue, load) its output tile at the same moment - HBM idles during the K loops and saturates during
    // the epilogues.  Workgroups that own one tile fewer than the fullest ones have a tile time of slack; they
    // spend a pseudo-random fraction of it BEFORE their first tile instead of after their last.
    {
        const int dunit = mode >> 8;                               // estimated cycles per K-tile, 0 = off
        const int slack = sc.slack();
        if (dunit > 0 && slack > 0) {
#ifndef HG_R2_DELAY_HASH
#define HG_R2_DELAY_HASH 1
#endif
            // what the pseudo-random fraction is drawn from: 1 (default) the workgroup's XCD - the slack workgroups of an XCD stay in
            // step with each other, they share activation panels through its L2: 583 instead of 670 MB fetched per launch, step
            // -0.8 % (profiles/r04_energy_ab5_stagger.txt); 0 the workgroup itself (rounds 1-3)
            const unsigned hkey = HG_R2_DELAY_HASH == 1 ? (unsigned)(bid & 7) * 37u + 11u : (unsigned)bid;
            const unsigned h = (hkey * 2654435761u) >> 24;   // 0..255
            const long long d = ((long long)slack * nk * dunit * h) >> 8;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < d) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int S = my_tiles * nk;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    // ---- DMA source offsets: a piece is 8 rows x 128 B; lane -> (row = l>>3, chunk' = l&7)
    int voffA[GA], voffW[GW];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + (lane >> 3);            // 0..127
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffA[i] = row * p.lda * 2 + c * 16 - i * 1024;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int row = (wave * GW + i) * 8 + (lane >> 3);            // 0..255 (W0 = 0..127, W1 = 128..255)
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffW[i] = row * p.K * 2 + c * 16 - i * 1024;
    }
    // ---- two load streams (A and W are issued in different phases)
    struct Ld { int kt, r, soff, st; };          // K-tile inside its tile, tile, tile origin (bytes), LDS stage offset
    Ld lA{nk - 1, -1, 0, (NST - 1) * STAGE}, lW{nk - 1, -1, 0, (NST - 1) * STAGE};
    // WRAP: 0 = the stream stays inside its tile, 1 = it moves to the next tile, 2 = decide at run time (prologue).
    // The consumer's K-tile position fixes it: the A stream (distance 2) wraps when K-tile nk-2 is consumed, the W
    // stream (distance 3) at K-tile nk-3.
    auto advance = [&](Ld& l, bool isA, auto WRAP_T) {
        constexpr int WRAP = decltype(WRAP_T)::value;
        ++l.kt;
        if (WRAP == 1 || (WRAP == 2 && l.kt == nk)) {
            l.kt = 0;
            ++l.r;
            int tm, tn;
            tile_of(l.r, tm, tn);
            l.soff = (xmode & 1) ? 0 : (isA ? tm * BM * p.lda * 2 : tn * 256 * p.K * 2);   // mode 1: every tile reads tile 0
        }
        l.st = l.st == (NST - 1) * STAGE ? 0 : l.st + STAGE;      // stage of stream position g is g % NST
    };
    // all pieces of a wave share one M0 (LDS base): piece i adds its 1 KiB through the instruction's immediate offset,
    // which the hardware also adds to the global address, so voff*[i] carry -1024 * i
    auto dma_A = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if (xmode & 64) return;   // timing experiment: no operand DMA
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + lA.st + wave * GA * 1024), 16, voffA[i],
                                                 lA.soff + lA.kt * (BK * 2), i * 1024, CON ? 16 /* sc1 */ : 0);
    };
    auto dma_W = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if (xmode & 64) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + lW.st + AB + wave * GW * 1024), 16, voffW[i],
                                                 lW.soff + lW.kt * (BK * 2), i * 1024, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;
    static_assert(GA == 2 && GW == 4, "piece helpers");
    auto issue_A = [&](auto WRAP_T) {
        advance(lA, true, WRAP_T);
        dma_A(P0{}); dma_A(P1{});
    };
    auto issue_W = [&](auto WRAP_T) {
        advance(lW, false, WRAP_T);
        dma_W(P0{}); dma_W(P1{}); dma_W(P2{}); dma_W(P3{});
    };
    using WDYN = std::integral_constant<int, 2>;

    // ---- fragment read offsets (row bases are multiples of 16 -> lane-constant swizzle)
    const int sw = (lane >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
    const int a_row = (wm * 32 + (lane & 15)) * 128;                 // + ha*8192 + f*2048
    const int w_row = AB + (wn * 32 + (lane & 15)) * 128;            // + hb*WH + g2*2048

    half8 xa[2][2], wb[2][2][2];
    auto read_A = [&](int ha, int st) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xa[f][ks] = *reinterpret_cast<const half8*>(smem + st + ha * 8192 + a_row + f * 2048 + coff[ks]);
    };
    auto read_W = [&](int st) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wb[hb][g2][ks] = *reinterpret_cast<const half8*>(smem + st + hb * WH + w_row + g2 * 2048 + coff[ks]);
    };
    f32x4 acc[2][2][2][2];
    auto mma = [&](auto HA) {
        constexpr int ha = decltype(HA)::value;
        if (xmode & 2) {   // timing experiment: no MFMAs (operands kept live)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int f = 0; f < 2; ++f) asm volatile("" ::"v"(xa[f][ks]));
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) asm volatile("" ::"v"(wb[hb][g2][ks]));
            }
            return;
        }
        SEG_B(3);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
                        acc[ha][hb][f][g2] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[hb][g2][ks], xa[f][ks], acc[ha][hb][f][g2], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEG_E(3);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto sync_fetch = [&]() {
        SEG_B(1);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); the builtin keeps the compiler's waitcnt scoreboard in sync
        SEG_E(1);
        SEG_B(2);
        barrier_raw();
        SEG_E(2);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        SEG_B(4);
        barrier_raw();
        SEG_E(4);
    };

    // ---- bias -> LDS once per workgroup
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < p.N / 4; i += 512)
            *reinterpret_cast<f32x4*>(smem + BIAS_OFF + i * 16) = p.bias ? reinterpret_cast<const f32x4*>(p.bias)[i] : z;
        if constexpr (GS)      // the next LayerNorm's weight behind the bias
            for (int i = tid; i < p.N / 4; i += 512)
                *reinterpret_cast<f32x4*>(smem + BIAS_OFF + p.N * 4 + i * 16) = reinterpret_cast<const f32x4*>(p.gamma)[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    if constexpr (CON) {      // the first tile's A panel must be complete before any of it is fetched
        if (wave == 0) sc.poll_blocking(0);
        barrier_raw();
    }
    // ---- prologue: W(0) A(0) W(1) A(1) W(2); A(2) is issued by the first PA, W(3) by the first PB
    issue_W(WDYN{}); issue_A(WDYN{});
    if (S > 1) { issue_W(WDYN{}); issue_A(WDYN{}); }
    if (S > 2) issue_W(WDYN{});
    if (S > 2) wait_vm<NWT>();
    else if (S > 1) wait_vm<GW + GA>();
    else wait_vm<0>();
    barrier_raw();
    const bool late = (wave >= 4) && !(xmode & 8);
    if (late) barrier_raw();

#ifdef HG_STAMPS
    t_all = __builtin_amdgcn_s_memtime();
#endif
    int stg = 0;                               // LDS stage of the current K-tile of the stream (position % NST)
    // the previous tile lay inside M, i.e. issued every one of its E epilogue stores (a ragged tile may skip store
    // instructions whose rows are all masked: the waits that follow it then do not allow for any)
    bool prev_full = false;
    unsigned polled = 0;                       // (CON, wave 0) the next tile's ready counter as read six K-tiles before the tile's end
    (void)polled;
    for (int r = 0; r < my_tiles; ++r) {
        int tm, tn;
        tile_of(r, tm, tn);
        const int m0 = tm * BM, n0 = tn * 256;
        const bool post_ok = prev_full;
        prev_full = m0 + BM <= p.M;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) acc[a][b][f][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 xres[RESID && !IN_HL ? 2 : 1][RESID && !IN_HL ? 2 : 1][RESID && !IN_HL ? 2 : 1][RESID && !IN_HL ? 2 : 1];
        float muv[RLN ? 2 : 1][RLN ? 2 : 1];          // EPI_RESID_LN: centre of this lane's rows for the fp16 copy
        typedef unsigned u32x4_hl __attribute__((ext_vector_type(4)));
        u32x4_hl xhi[IN_HL ? 2 : 1][IN_HL ? 2 : 1][IN_HL ? 2 : 1], xlo[IN_HL ? 2 : 1][IN_HL ? 2 : 1][IN_HL ? 2 : 1];   // [ha][hb][g2]
        float mucv[IN_HL ? 2 : 1][IN_HL ? 2 : 1];     // centre the hi / lo being read were written with
        // lo piece (ha, hb, g2) of this wave and tile: 64 lanes x 16 B
        auto lo_ptr = [&](int tm_, int tn_, int ha, int hb, int g2) {
            return p.lo + ((((size_t)tm_ * tiles_n + tn_) * 8 + wave) * 8 + (ha * 4 + hb * 2 + g2)) * (HG_LO8 ? 256 : 512) +
                   lane * (HG_LO8 ? 4 : 8);
        };
        // One K-tile.  KIND: 0 middle, 1 first of a tile (the previous epilogue's stores may be pending), 2 / 3 / 4 the
        // third-to-last, second-to-last and last K-tile of a tile: only there the refills (A at distance 2, W at
        // distance 3) and the waits depend on whether another tile follows.  K >= 256 keeps the kinds distinct.
        // (CON) 5 / 6: K-tiles nk - 6 / nk - 5, middle K-tiles in which wave 0 issues the poll of the next tile's panel / still has
        // it among the NWT + 1 youngest operations of its stream; kind 2 then starts by looking at the polled value
        auto ktile = [&](auto KIND_T) {
            constexpr int KINDX = decltype(KIND_T)::value;
            constexpr int KIND = KINDX >= 5 ? 0 : KINDX;
            const int st = stg * STAGE;
            stg = stg == NST - 1 ? 0 : stg + 1;
            const bool more = KIND < 2 || r + 1 < my_tiles;
            const bool pollw = CON && wave == 0 && r + 1 < my_tiles;      // this wave polls for a next tile
            (void)more; (void)pollw;
            if constexpr (CON && KIND == 2) {
                if (pollw) sc.poll_finish(r + 1, polled);
            }
            // ---------------- PA: fetch A0, W0, W1 of this K-tile; refill A(g+2); quadrants (A0,W0) (A0,W1)
            read_A(0, st);
            read_W(st);
            if (KIND < 3 || more) issue_A(std::integral_constant<int, KIND == 3 ? 1 : 0>{});   // K-tile g+2 exists
            if constexpr (RESID && KIND == 4) {
                if constexpr (IN_HL) {      // hi (paired 16-byte pieces of the row-major copy), lo (this wave's own pieces), centres
                    const int qq = lane >> 4;
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha) {
                        int mp = m0 + ha * 64 + wm * 32 + (lane & 15) + ((qq & 1) ? 16 : 0);
                        mp = mp < p.M ? mp : p.M - 1;
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2) {
                                xhi[ha][hb][g2] = *reinterpret_cast<const u32x4_hl*>(
                                    p.out2 + (size_t)mp * p.ld2 + n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * (qq & ~1));
                                if constexpr (HG_LO8) {
                                    typedef unsigned u32x2_hl __attribute__((ext_vector_type(2)));
                                    const u32x2_hl l8 = *reinterpret_cast<const u32x2_hl*>(lo_ptr(tm, tn, ha, hb, g2));
                                    xlo[ha][hb][g2] = u32x4_hl{l8[0], l8[1], 0u, 0u};
                                } else {
                                    xlo[ha][hb][g2] = *reinterpret_cast<const u32x4_hl*>(lo_ptr(tm, tn, ha, hb, g2));
                                }
                            }
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            int m = m0 + ha * 64 + wm * 32 + f * 16 + (lane & 15);
                            m = m < p.M ? m : p.M - 1;
                            muv[ha][f] = p.mu[m];
                            mucv[ha][f] = p.muc[m];
                        }
                    }
                } else {      // residual rows of this tile, needed by the epilogue one K-tile later
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha)
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
                                    xres[ha][hb][f][g2] = *reinterpret_cast<const f32x4*>(
                                        reinterpret_cast<const float*>(p.out) + (size_t)m * p.ldc + n);
                                }
                        }
                }
            }
            sync_fetch();
            mma(I0{});
            sync_mma();
            // ---------------- PB: fetch A1; refill W(g+3); wait for A,W(g+1); quadrants (A1,W0) (A1,W1)
            read_A(1, st);
            if (KIND < 2 || more) issue_W(std::integral_constant<int, KIND == 2 ? 1 : 0>{});   // K-tile g+3 exists
            if constexpr (KINDX == 5) {
                if (pollw) polled = sc.poll_issue(r + 1);
            }
            SEG_B(0);
            if constexpr (KINDX == 5 || KINDX == 6) { if (pollw) wait_vm<NWT + 1>(); else wait_vm<NWT>(); }
            else if constexpr (KIND == 0) wait_vm<NWT>();
            else if constexpr (KIND == 1) { if (post_ok) wait_vm<NWT + E>(); else wait_vm<NWT>(); }
            else if constexpr (KIND == 4) { if (more) wait_vm<NWT + R>(); }       // no successor: nothing to wait for
            else { if (more) wait_vm<NWT>(); else wait_vm<0>(); }
            SEG_E(0);
            sync_fetch();
            mma(I1{});
            sync_mma();
        };
        {
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
            using K3 = std::integral_constant<int, 3>;
            using K4 = std::integral_constant<int, 4>;
            ktile(K1{});
            if constexpr (CON) {
                for (int kt = 1; kt < nk - 6; ++kt) ktile(K0{});
                ktile(std::integral_constant<int, 5>{});
                ktile(std::integral_constant<int, 6>{});
                ktile(K0{});
            } else {
                for (int kt = 1; kt < nk - 3; ++kt) ktile(K0{});
            }
            ktile(K2{});
            ktile(K3{});
            ktile(K4{});
        }
        // ---------------- epilogue
        SEG_B(7);
        if (xmode & 4) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) asm volatile("" ::"v"(acc[a][b][f][g2]));
            continue;
        }
        const int q = lane >> 4;
        // tiles entirely inside M (all of them at M = 197 * 256) skip the per-store row masks
        auto epilogue = [&](auto INTERIOR_T) {
        constexpr bool INTERIOR = decltype(INTERIOR_T)::value;
        if constexpr (F16OUT) {
            half_t* outp = reinterpret_cast<half_t*>(p.out);
#pragma unroll
            for (int ha = 0; ha < 2; ++ha) {
                const int mX = m0 + ha * 64 + wm * 32 + (lane & 15);
                const int m = mX + ((q & 1) ? 16 : 0);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int nb = n0 + hb * 128 + wn * 32 + g2 * 16;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (nb + 4 * q) * 4);
                        f32x4 vx = acc[ha][hb][0][g2] + bv, vy = acc[ha][hb][1][g2] + bv;
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
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x2 ux = __builtin_bit_cast(u32x2, hx), uy = __builtin_bit_cast(u32x2, hy);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        if (INTERIOR || m < p.M) *reinterpret_cast<u32x4*>(outp + (size_t)m * p.ldc + nb + 4 * (q & ~1)) = o;
                    }
            }
        } else if constexpr (RLN) {
            // x' = x + acc + bias ; x16 = fp16(x' - mu[row]) with mu = the row's previous mean (keeps the fp16 rounding
            // relative to the row's spread, not to its offset) ; per row and per wave column group (64 columns) the pair
            // (sum, sum of squared deviations from the group mean) for the next LayerNorm.  The stream itself: fp32 in
            // place, or (HL) centre + hi + lo with hi = that very copy and lo = fp16((x' - mu) - hi).
            half_t* out2 = p.out2;
            const int sg = tn * 4 + wn;                     // column group of this wave
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int ha = 0; ha < 2; ++ha) {
                half4 h16[2][2][2];                         // [f][hb][g2]: the new copy (hi)
                half4 g16[GS && OUT_HL ? 2 : 1][GS && OUT_HL ? 2 : 1][GS && OUT_HL ? 2 : 1];      // ... times gamma, where hi must stay unscaled
                half4 l16[OUT_HL ? 2 : 1][OUT_HL ? 2 : 1][OUT_HL ? 2 : 1];
                unsigned l8[OUT_HL ? 2 : 1][OUT_HL ? 2 : 1][OUT_HL ? 2 : 1];      // HG_LO8: the remainder as four bf8 (e5m2)
                half4 hin[IN_HL ? 2 : 1][IN_HL ? 2 : 1][IN_HL ? 2 : 1];
                f32x4 lin[IN_HL ? 2 : 1][IN_HL ? 2 : 1][IN_HL ? 2 : 1];
                if constexpr (IN_HL) {
                    // the copy was stored with the row tiles f = 0, 1 paired through v_permlane16_swap (below); the same
                    // exchange gives every lane its own two row tiles back; the lo piece holds them side by side
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) {
                            const u32x4_hl o = xhi[ha][hb][g2], l = xlo[ha][hb][g2];
                            const auto s0 = __builtin_amdgcn_permlane16_swap(o[0], o[2], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(o[1], o[3], false, false);
                            hin[0][hb][g2] = __builtin_bit_cast(half4, u32x2{(unsigned)s0[0], (unsigned)s1[0]});
                            hin[1][hb][g2] = __builtin_bit_cast(half4, u32x2{(unsigned)s0[1], (unsigned)s1[1]});
                            if constexpr (HG_LO8) {
#pragma unroll
                                for (int f = 0; f < 2; ++f) {
                                    const auto a = __builtin_amdgcn_cvt_pk_f32_bf8((int)l[f], false);
                                    const auto b = __builtin_amdgcn_cvt_pk_f32_bf8((int)l[f], true);
                                    lin[f][hb][g2] = f32x4{a[0], a[1], b[0], b[1]};
                                }
                            } else {
#pragma unroll
                                for (int f = 0; f < 2; ++f) {
                                    const half4 lh = __builtin_bit_cast(half4, u32x2{l[2 * f], l[2 * f + 1]});
                                    lin[f][hb][g2] = f32x4{(float)lh[0], (float)lh[1], (float)lh[2], (float)lh[3]};
                                }
                            }
                        }
                }
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
                            f32x4 xin;
                            if constexpr (IN_HL) {      // (centre + hi) + lo, two elements per instruction
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
                                const f32x2 mc2 = {mucv[ha][f], mucv[ha][f]};
#pragma unroll
                                for (int e2 = 0; e2 < 2; ++e2) {
                                    const half2v h2 = {hin[f][hb][g2][2 * e2], hin[f][hb][g2][2 * e2 + 1]};
                                    // (lo is stored scaled by HG_LO_SCALE when it is bf8: one packed fma instead of the add)
                                    const f32x2 ls2 = {HG_LO8 ? 1.0f / HG_LO_SCALE : 1.0f, HG_LO8 ? 1.0f / HG_LO_SCALE : 1.0f};
                                    const f32x2 x2 = f32x2{lin[f][hb][g2][2 * e2], lin[f][hb][g2][2 * e2 + 1]} * ls2 +
                                                     (mc2 + __builtin_convertvector(h2, f32x2));
                                    xin[2 * e2] = x2[0];
                                    xin[2 * e2 + 1] = x2[1];
                                }
                            } else {
                                xin = xres[ha][hb][f][g2];
                            }
                            v[hb][g2] = xin + (acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4));
                            sum += (v[hb][g2][0] + v[hb][g2][1]) + (v[hb][g2][2] + v[hb][g2][3]);
                            if constexpr (!OUT_HL) {
                                if (INTERIOR || m < p.M)
                                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = v[hb][g2];
                            }
                            // (two elements per instruction: v_pk_add_f32, v_cvt_pk_f16_f32 - RNE like the scalar conversion; the copy is
                            // converted once and read back from its packed form)
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            const f32x2 mu2 = {muv[ha][f], muv[ha][f]};
                            f32x4 gm4 = f32x4{1.f, 1.f, 1.f, 1.f};
                            if constexpr (GS) gm4 = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + p.N * 4 + n * 4);      // (one 16-byte read per four columns)
                            float rem[4];
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2) {
                                const f32x2 d = f32x2{v[hb][g2][2 * e2], v[hb][g2][2 * e2 + 1]} - mu2;
                                half2v hh = __builtin_convertvector(d, half2v);
                                if constexpr (GS) {
                                    const f32x2 gm2 = f32x2{gm4[2 * e2], gm4[2 * e2 + 1]};
                                    const half2v hg = __builtin_convertvector(d * gm2, half2v);
                                    if constexpr (OUT_HL) {
                                        g16[f][hb][g2][2 * e2] = hg[0];
                                        g16[f][hb][g2][2 * e2 + 1] = hg[1];
                                    } else {
                                        hh = hg;      // (the stream leaves as fp32: the copy has no second role)
                                    }
                                }
                                h16[f][hb][g2][2 * e2] = hh[0];
                                h16[f][hb][g2][2 * e2 + 1] = hh[1];
                                if constexpr (OUT_HL) {
                                    const f32x2 r = d - __builtin_convertvector(hh, f32x2);
                                    const f32x2 rs = HG_LO8 ? r * f32x2{HG_LO_SCALE, HG_LO_SCALE} : r;
                                    rem[2 * e2] = rs[0];
                                    rem[2 * e2 + 1] = rs[1];
                                    if constexpr (!HG_LO8) {
                                        l16[f][hb][g2][2 * e2] = (half_t)r[0];
                                        l16[f][hb][g2][2 * e2 + 1] = (half_t)r[1];
                                    }
                                }
                            }
                            if constexpr (OUT_HL && HG_LO8) {
                                int w8 = __builtin_amdgcn_cvt_pk_bf8_f32(rem[0], rem[1], 0, false);
                                w8 = __builtin_amdgcn_cvt_pk_bf8_f32(rem[2], rem[3], w8, true);
                                l8[f][hb][g2] = (unsigned)w8;
                            }
                        }
                    sum = sum_rows(sum);
                    const float gm = sum * (1.0f / 64.0f);
                    float m2 = 0.f;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2) {      // (subtraction in pairs; the sum keeps its order)
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
                                const f32x2 d = f32x2{v[hb][g2][2 * e2], v[hb][g2][2 * e2 + 1]} - f32x2{gm, gm};
                                m2 = fmaf(d[0], d[0], m2);
                                m2 = fmaf(d[1], d[1], m2);
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
                        const u32x2 ux = __builtin_bit_cast(u32x2, h16[0][hb][g2]), uy = __builtin_bit_cast(u32x2, h16[1][hb][g2]);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        if (INTERIOR || m < p.M) *reinterpret_cast<u32x4*>(out2 + (size_t)m * p.ld2 + nb + 4 * (q & ~1)) = o;
                        if constexpr (GS && OUT_HL) {      // the scaled copy beside the stream's hi half
                            const u32x2 gx = __builtin_bit_cast(u32x2, g16[0][hb][g2]), gy = __builtin_bit_cast(u32x2, g16[1][hb][g2]);
                            const auto t0 = __builtin_amdgcn_permlane16_swap(gx[0], gy[0], false, false);
                            const auto t1 = __builtin_amdgcn_permlane16_swap(gx[1], gy[1], false, false);
                            const u32x4 og = {t0[0], t1[0], t0[1], t1[1]};
                            if (INTERIOR || m < p.M) *reinterpret_cast<u32x4*>(p.out3 + (size_t)m * p.ld3 + nb + 4 * (q & ~1)) = og;
                        }
                        if constexpr (OUT_HL) {      // the remainder: this lane's two row tiles side by side, the wave's piece contiguous
                            if constexpr (HG_LO8) {
                                *reinterpret_cast<u32x2*>(lo_ptr(tm, tn, ha, hb, g2)) = u32x2{l8[0][hb][g2], l8[1][hb][g2]};
                            } else {
                                const u32x2 lx = __builtin_bit_cast(u32x2, l16[0][hb][g2]), ly = __builtin_bit_cast(u32x2, l16[1][hb][g2]);
                                *reinterpret_cast<u32x4*>(lo_ptr(tm, tn, ha, hb, g2)) = u32x4{lx[0], lx[1], ly[0], ly[1]};
                            }
                        }
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
                            f32x4 v = acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4);
                            if constexpr (RESID) {
                                if (INTERIOR || m < p.M) {
                                    if constexpr (EPI == EPI_SCALE_RESID_F32) v *= *reinterpret_cast<const f32x4*>(p.pos + n);
                                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) =
                                        xres[ha][hb][f][g2] + v;
                                }
                            } else {
                                epilogue_ring<EPI>(p, m, n, v);
                            }
                        }
                }
        }
        };
        if (m0 + BM <= p.M) epilogue(std::true_type{});
        else epilogue(std::false_type{});
        SEG_E(7);
    }
#ifdef HG_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 8 + wave) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = tacc[k];
        d[8] = __builtin_amdgcn_s_memtime() - t_all;
        d[9] = (unsigned long long)my_tiles * nk;
    }
#endif
    if (!late) barrier_raw();   // balances the extra barrier of the late waves
#endif
}

}  // namespace hg
