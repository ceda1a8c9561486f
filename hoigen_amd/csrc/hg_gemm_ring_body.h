// Body of the persistent ring GEMM (hg_gemm_ring.hip holds the kernel's description, the stand-alone kernel and its launchers) as a
// device function over a tile schedule, so that the MLP pair kernel (hg_mlp_pair.hip: c_fc -> QuickGELU -> c_proj in ONE launch) runs
// the very same K loop and epilogues on its own tile order.  SCHED: n_items(), tile(r, tm, tn), slack(), PUBLISH (+ publish(tm, lane)).
#pragma once
#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

// Diagnostic build (HG_EXTRA_FLAGS="-DHG_STAMPS -DHG_STAMP_MASK=<bits>"): per-wave s_memtime totals of the
// selected segment kinds (0 vmcnt waits, 1 lgkmcnt waits, 2 fetch barriers, 3 MFMA segments, 4 MFMA barriers,
// 5 DMA issue, 6 ds_read issue, 7 epilogue) and of the whole tile loop, written to GemmArgs::dbg.
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

template <int MF, int EPI, bool PH2, class SCHED>
__device__ __forceinline__ void gemm_ring_body(const GemmArgs& p, const unsigned a_bytes, const int mode, const SCHED& sc) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtins (buffer resources, LDS DMA): host sees just the stub
    // timing-experiment switches (HG_RING_MODE bits 1 locality, 2 no MFMA, 4 no epilogue, 8 no stagger, 32 no fragment
    // reads, 64 no operand DMA) exist only in a -DHG_EXPERIMENTS build: run-time branches in the K loop cost several per
    // cent.  (The store experiments of round 2 - junk stores trickled under the next tile, epilogue without stores - are in
    // the history: commits "Ring2 junk-trickle experiment", "experiment: phase-shifted half tiles"; results in DESIGN.md 4.)
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int BM = 64 * MF, BK = 64;
    constexpr int AH = MF * 4096, BH = 16384;          // bytes per A / W half-tile slot
    constexpr int STAGE = 2 * AH + 2 * BH;
    constexpr int GA = MF / 2, GB = 2;
    static_assert(GA <= 2, "DMA piece helpers cover two pieces per half-tile");
    constexpr int N1 = 2 * GA + 3 * GB, N2 = 3 * GA + 2 * GB;
    constexpr bool RLN = (EPI == EPI_RESID_LN_F32);    // residual + centred fp16 copy + LayerNorm statistics (MF = 4)
    constexpr bool RESID = (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_SCALE_RESID_F32 || RLN);
    // epilogue store instructions per wave (vmcnt immediates are 6 bits: anything above is clamped, i.e. stricter)
    // (fp16 outputs leave as paired 16-byte stores: half as many; counting too many here would let the first waits of
    // the next tile pass before its operands have landed)
    constexpr bool F16_STORES = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16 ||
                                 EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_QGELU_F16);
    constexpr int E_RAW = RLN ? 8 * MF + 4 * MF + 2 * MF : (F16_STORES ? 4 * MF : 8 * MF);
    constexpr int E = E_RAW > 52 ? 52 : E_RAW;
    // RESID at MF = 2: the residual rows are fetched one K-tile before the epilogue (64 spare VGPRs)
    constexpr bool XPRE = RESID && MF == 2;
    // folded LayerNorm (EPI_LN_*): (mean, rstd) of this lane's 2*MF rows are fetched one K-tile ahead as well
    constexpr bool LNC = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_QGELU_F16);
    // RESID at MF = 4: no room for all 32 residual chunks; a rolling window of ROLL_W chunks (f32x4 per lane) is
    // filled one K-tile before the epilogue and refilled as the epilogue consumes it (the fragment registers are
    // dead by then, the accumulators die chunk by chunk)
    constexpr bool ROLL = RESID && MF == 4;
    constexpr int ROLL_W = RLN ? 6 : 8;               // 8 with the LayerNorm extras spills 3 VGPRs
    constexpr int R = XPRE ? E : (ROLL ? ROLL_W + (RLN ? 1 : 0) : 0);   // prefetch loads per wave in the last K-tile
    // PUB (SCHED::PUBLISH, the MLP pair kernel's c_fc): the output tile is handed to OTHER workgroups of this launch, all of them ON THE
    // SAME XCD (the schedule deals a row panel's producers and consumers to one hardware XCC id, hg_mlp_pair.hip): plain stores.  Every
    // wave's stores of tile r have retired, i.e. are in that XCD's L2, once it has passed the counted wait of the first phase of the
    // second K-tile of tile r + 1 (ten operations issued behind the stores, at most eight outstanding); the two barriers of that phase
    // tell wave 0 that all eight waves have (waves 4-7 run one barrier behind), and wave 0 then adds 1 to the row panel's ready counter
    // (sc.publish) - no drain of the operand ring, one atomic per tile.  Behind the last tile of a segment: drain, barrier, publish.
    // The atomic is one more operation in wave 0's vmcnt stream: K-tile kinds 4 / 5 (the second and third K-tile of a tile) run the two
    // waits it is younger than with one more allowed.  (Per-wave atomics - 96 per panel - cost 14 us per launch; stored write-through
    // (sc1) for consumers behind another L2 the epilogue costs 50 us per launch: profiles/r06_mlp_pair.txt.)
    constexpr bool PUB = SCHED::PUBLISH;
    static_assert(!PUB || (PH2 && (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_QGELU_F16)), "hand-off: two-phase loop, fp16 LayerNorm-fold epilogues");
    constexpr int BIAS_OFF = 2 * STAGE;                // bias[N] (and cs[N] for EPI_LN_*) staged in LDS behind the ring
    const int MR_OFF = BIAS_OFF + 2 * p.N * 4;         // EPI_LN_*: (mean, rstd) of the tile's BM rows, 8 B each
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef HG_STAMPS
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_beg = 0, t_all = 0;
#endif
    // HG_TRACE build: time stamps (s_memtime) of one wave around the boundary between its second and third tile: after
    // the last K-tiles of tile 1, after its epilogue, after the first K-tiles of tile 2 (tools/gpu_ring_trace.sh)
#ifdef HG_TRACE
    unsigned long long ttr[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int ttn = 0;
#define HG_TR(cond) do { if ((cond) && ttn < 16) { ttr[ttn] = __builtin_amdgcn_s_memtime(); ++ttn; } } while (0)
#else
#define HG_TR(cond) do {} while (0)
#endif

    // (the MLP pair kernel's schedules rebuild the thread id from the wave index and v_mbcnt, behind an opaque move: threadIdx.x itself
    // would have to stay in v0 across the other body, and with v0 / v1 taken every register tuple of this body - accumulators, fragments -
    // starts at 2 (mod 4) instead of 0: the same K loop then runs 6 % slower (profiles/r06_mlp_pair.txt))
    const int tid = sc.thread_id(), lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    // ---- this workgroup's tiles come from the schedule `sc`: RingTileList (hg_gemm_dev.h) for the stand-alone kernels - XCD-contiguous
    // chunks of an n-group-major list, tiles slot, slot + cpx, ... -; the MLP pair kernel (hg_mlp_pair.hip) deals row-panel-progressive
    // lists and hands finished row panels to its c_proj tiles (SCHED::PUBLISH)
    const int bid = blockIdx.x;
    const int n_items = sc.n_items();
    const int my_tiles = n_items;
    const int tiles_m_all = (p.M + BM - 1) / BM;
    // item e of this workgroup: its tile (every item sweeps all of K)
    auto item_get = [&](int e, int& tm, int& tn, int& kb, int& ke) {
        kb = 0;
        ke = nk;
        sc.tile(e, tm, tn);
    };
    // De-synchronised epilogues: all tiles take the same time, so every CU would store (and, for the residual
    // epilogue, load) its output tile at the same moment - HBM idles during the K loops and saturates during
    // the epilogues.  Workgroups that own one tile fewer than the fullest ones have a tile time of slack; they
    // spend a pseudo-random fraction of it BEFORE their first tile instead of after their last.
    {
        const int dunit = mode >> 8;                               // estimated cycles per K-tile, 0 = off
        const int slack = sc.slack();                              // tiles fewer than the fullest workgroups own
        if (dunit > 0 && slack > 0) {
            const unsigned h = ((unsigned)bid * 2654435761u) >> 24;   // 0..255
            const long long d = ((long long)slack * nk * dunit * h) >> 8;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < d) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int S = my_tiles * nk;                   // K-tiles in this workgroup's stream

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.mr, 0, LNC ? (unsigned)((size_t)tiles_m_all * BM * 8) : 0u, 0x00020000);
    // ---- DMA source offsets (bytes, per lane; identical for every tile and K-step)
    int voffA[2][GA], voffW[2][GB];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            const int row = (wave * GA + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            voffA[h][i] = (h * (BM / 2) + row) * p.lda * 2 + c * 16 - i * 1024;
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            const int row = (wave * GB + i) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            // fp16-output kernels: LDS row `row` of W half h holds output column (row/32)*64 + 16*((row%16)/4) +
            // 4*(2h + (row%32)/16) + row%4 of the tile, so that wave wn owns the 64 CONSECUTIVE columns wn*64.. and a
            // lane's four MFMA blocks (hb, g2) hold 16 consecutive ones: rows leave as whole 128-byte lines (epilogue)
            const int src = F16_STORES ? (row >> 5) * 64 + 16 * ((row & 15) >> 2) + 4 * (2 * h + ((row & 31) >> 4)) + (row & 3)
                                       : h * 128 + row;
            voffW[h][i] = src * p.K * 2 + c * 16 - i * 1024;
        }
    }
    // ---- load-stream state (wave-uniform): position ld_g, its tile origin and K offset
    int ld_g = -1, ld_kt = nk - 1, ld_ke = nk, ld_r = -1, ld_sA = 0, ld_sW = 0, ld_buf = 0;
    // WRAP: 0 = the stream stays inside its tile, 1 = it moves to the next tile, 2 = decide at run time.  In the
    // two-phase loop the stream (two K-tiles ahead) wraps exactly when K-tile nk-2 is consumed.
    auto ld_advance = [&](auto WRAP_T) {
        constexpr int WRAP = decltype(WRAP_T)::value;
        ++ld_g;
        ++ld_kt;
        if (WRAP == 1 || (WRAP == 2 && ld_kt == ld_ke)) {
            ++ld_r;
            int tm, tn;
            item_get(ld_r, tm, tn, ld_kt, ld_ke);
            ld_sA = (xmode & 1) ? 0 : tm * BM * p.lda * 2;     // mode 1 (timing experiment): every tile reads tile 0
            ld_sW = (xmode & 1) ? 0 : tn * 256 * p.K * 2;
        }
        ld_buf = (ld_g & 1) * STAGE;
    };
    // pieces [i0, i1) of a half-tile (interleaving the pieces with the segment's LDS reads was measured: no
    // gain).  All pieces of a half-tile share one M0 (LDS base): piece i adds its
    // 1 KiB through the instruction's immediate offset, which the hardware also adds to the global address,
    // so voff*[h][i] carry -1024 * i
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    auto dma_A = [&](int h, auto I) {
        constexpr int i = decltype(I)::value;
        if (xmode & 64) return;   // timing experiment: no operand DMA
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + ld_buf + h * AH + wave * GA * 1024), 16,
                                                 voffA[h][i], ld_sA + ld_kt * (BK * 2), i * 1024, 0);
    };
    auto dma_W = [&](int h, auto I) {
        constexpr int i = decltype(I)::value;
        if (xmode & 64) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + ld_buf + 2 * AH + h * BH + wave * GB * 1024),
                                                 16, voffW[h][i], ld_sW + ld_kt * (BK * 2), i * 1024, 0);
    };
    auto issue_A = [&](int h, int i0, int i1) {
        if (i0 <= 0 && 0 < i1) dma_A(h, P0{});
        if constexpr (GA > 1) {
            if (i0 <= 1 && 1 < i1) dma_A(h, P1{});
        }
    };
    auto issue_W = [&](int h, int i0, int i1) {
        if (i0 <= 0 && 0 < i1) dma_W(h, P0{});
        if (i0 <= 1 && 1 < i1) dma_W(h, P1{});
    };

    // ---- fragment read offsets
    const int sw = (lane >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
    const int a_row = (wm * MF * 16 + (lane & 15)) * 128;
    const int w_row = 2 * AH + (wn * 32 + (lane & 15)) * 128;

    // Fragment registers: one A set (half 0 in P1-P2, half 1 in P3-P4) and both W halves.
    half8 xa[MF][2], wb[2][2][2];
#ifdef HG_EXPERIMENTS
    if (xmode & 32) {   // defined (opaque) fragment values for the no-read experiment
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int f = 0; f < MF; ++f) asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "=v"(((int*)&xa[f][ks])[0]), "=v"(((int*)&xa[f][ks])[1]), "=v"(((int*)&xa[f][ks])[2]), "=v"(((int*)&xa[f][ks])[3]));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "=v"(((int*)&wb[h][g2][ks])[0]), "=v"(((int*)&wb[h][g2][ks])[1]), "=v"(((int*)&wb[h][g2][ks])[2]), "=v"(((int*)&wb[h][g2][ks])[3]));
        }
    }
#endif
    auto read_A = [&](int h, int buf) {
        if (xmode & 32) return;   // timing experiment: no fragment reads
#pragma unroll
        for (int f = 0; f < MF; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xa[f][ks] = *reinterpret_cast<const half8*>(smem + buf + h * AH + a_row + f * 2048 + coff[ks]);
    };
    auto read_W = [&](auto H, int buf) {
        constexpr int h = decltype(H)::value;
        if (xmode & 32) return;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wb[h][g2][ks] = *reinterpret_cast<const half8*>(smem + buf + h * BH + w_row + g2 * 2048 + coff[ks]);
    };

    f32x4 acc[2][2][MF][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < MF; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) acc[a][b][f][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma = [&](auto HA, auto HB) {
        constexpr int ha = decltype(HA)::value, hb = decltype(HB)::value;
        if (xmode & 2) {   // timing experiment: no MFMAs (operands kept live)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int f = 0; f < MF; ++f) asm volatile("" ::"v"(xa[f][ks]));
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
            for (int f = 0; f < MF; ++f)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
                    acc[ha][hb][f][g2] =
                        __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[hb][g2][ks], xa[f][ks], acc[ha][hb][f][g2], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        SEG_E(3);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // end of a fetch segment: this wave's fragment reads are complete, then the workgroup barrier; the
    // sched_barrier keeps the compiler from hoisting the (register-only) MFMAs into the fetch segment
    auto sync_fetch = [&]() {
        // the builtin (not inline asm) so that the compiler's waitcnt pass knows the LDS reads have returned:
        // with an opaque asm it keeps them on its scoreboard and throttles the next segment's ds_reads
        SEG_B(1);
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
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

    // ---- bias -> LDS once per workgroup (epilogue reads must not touch vmcnt: a register-returning
    // global load would wait for every older DMA of the ring)
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < p.N / 4; i += 512) {
            *reinterpret_cast<f32x4*>(smem + BIAS_OFF + i * 16) = p.bias ? reinterpret_cast<const f32x4*>(p.bias)[i] : z;
            if constexpr (LNC)
                *reinterpret_cast<f32x4*>(smem + BIAS_OFF + p.N * 4 + i * 16) = reinterpret_cast<const f32x4*>(p.cs)[i];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // ---- prologue: stream positions 0 and 1 (A1 of position 1 is issued in the first P1)
    ld_advance(std::integral_constant<int, 2>{});
    issue_A(0, 0, GA); issue_W(0, 0, GB); issue_W(1, 0, GB); issue_A(1, 0, GA);
    if (S > 1) {
        ld_advance(std::integral_constant<int, 2>{});
        issue_A(0, 0, GA); issue_W(0, 0, GB); issue_W(1, 0, GB);
        if constexpr (PH2) wait_vm<2 * GA + 2 * GB>();   // A0, W0, W1 of position 0 landed
        else wait_vm<N1>();                // A0, W0 of position 0 landed
    } else {
        if constexpr (PH2) wait_vm<GA>();
        else wait_vm<GA + GB>();
    }
    barrier_raw();
    // Stagger: waves 4-7 (the second wave of every SIMD) run one barrier interval behind waves 0-3, so a
    // SIMD always has one wave in a fetch segment (LDS reads, DMA issue, waits) and one in an MFMA segment.
    const bool late = (wave >= 4) && !(xmode & 8);
    if (late) barrier_raw();

#ifdef HG_STAMPS
    t_all = __builtin_amdgcn_s_memtime();
#endif
    int g = 0;
    // the previous tile lay inside M, i.e. issued every one of its E epilogue stores (a ragged tile may skip store
    // instructions whose rows are all masked: the waits that follow it then do not allow for any)
    bool prev_full = false;
    int tm_prev = 0;                               // (PUB) row panel of the previous tile: published in this tile's second K-tile
    (void)tm_prev;
    for (int r = 0; r < n_items; ++r) {
        int tm, tn, kb_r, ke_r;
        item_get(r, tm, tn, kb_r, ke_r);
        const int klen = nk;
        const int m0 = tm * BM, n0 = tn * 256;
        const bool post_ok = prev_full;
        prev_full = m0 + BM <= p.M;
        zero_acc();
        f32x4 xres[XPRE ? 2 : 1][XPRE ? 2 : 1][XPRE ? MF : 1][XPRE ? 2 : 1];
        // rolling residual window (ROLL): chunk c = ((ha * MF + f) * 2 + hb) * 2 + g2 is this lane's f32x4 of row
        // m0 + ha*BM/2 + wm*MF*16 + f*16 + (lane&15), columns n0 + hb*128 + wn*32 + g2*16 + 4*(lane>>4)
        f32x4 xw[ROLL ? ROLL_W : 1];
        float muw[RLN ? 2 : 1];
        auto chunk_row = [&](int rg) {
            int m = m0 + (rg / MF) * (BM / 2) + wm * MF * 16 + (rg % MF) * 16 + (lane & 15);
            return m < p.M ? m : p.M - 1;
        };
        auto chunk_load = [&](int c) {
            const int rg = c >> 2, hb = (c >> 1) & 1, g2 = c & 1;
            const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * (lane >> 4);
            return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.out) + (size_t)chunk_row(rg) * p.ldc + n);
        };
        // One K-tile of the two-phase schedule.  KIND: 0 middle, 1 first of a tile (the previous epilogue's stores may be
        // pending), 2 second to last (LayerNorm statistics DMA), 3 last (residual prefetch).  K >= 256 makes the four
        // kinds distinct K-tiles, and every K-tile before the last two of a tile has two successors in the stream, so
        // the middle of the loop carries no run-time conditions at all.
        // (PUB) 4 / 5: the second / third K-tile of a tile - middle K-tiles that publish the previous tile (`pub`: there is one) / still
        // have its atomic among the NP + 1 youngest operations of the stream
        auto ph2_ktile = [&](auto KIND_T) {
            constexpr int KINDX = decltype(KIND_T)::value;
            constexpr int KIND = KINDX >= 4 ? 0 : KINDX;
            const int buf = (g & 1) * STAGE;
            const bool post = post_ok;
            const bool more = KIND < 2 || r + 1 < n_items;       // a K-tile two positions ahead exists
            const bool pub = PUB && r > 0 && wave == 0;      // this wave publishes the previous tile
            (void)post; (void)more; (void)pub;
            // Two phases per K-tile (32 MFMAs per segment, half the barriers):
            //   PA: fetch A0 W0 W1 (t); refill A1(t+1);               wait -> A1(t) landed;        quadrants (0,0) (0,1)
            //   PB: fetch A1 (t);       refill A0 W0 W1 (t+2);         wait -> A0 W0 W1 (t+1) landed; quadrants (1,1) (1,0)
            // both waits leave NP = 2GA+2GB DMA instructions (64 KiB per CU) in flight
            constexpr int NP = 2 * GA + 2 * GB;
            read_A(0, buf);
            read_W(I0{}, buf);
            read_W(I1{}, buf);
            if constexpr (LNC && KIND == 2) {
                if (lane < BM / 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + MR_OFF + wave * BM), 16, lane * 16,
                                                             (m0 + wave * (BM / 8)) * 8, 0, 0);
            }
            if (KIND < 3 || more) issue_A(1, 0, GA);      // A1 of position g+1 exists unless the stream ends here
            if constexpr (ROLL && KIND == 3) {
                {      // first ROLL_W residual chunks (+ the first row group's centre) of this tile
#pragma unroll
                    for (int c = 0; c < ROLL_W; ++c) xw[c] = chunk_load(c);
                    if constexpr (RLN) muw[0] = p.mu[chunk_row(0)];
                }
            }
            if constexpr (XPRE && KIND == 3) {
                {
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                        for (int f = 0; f < MF; ++f) {
                            int m = m0 + ha * (BM / 2) + wm * MF * 16 + f * 16 + (lane & 15);
                            m = m < p.M ? m : p.M - 1;
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
            SEG_B(0);
            if constexpr (KINDX == 5) { if (pub) wait_vm<NP + 1>(); else wait_vm<NP>(); }
            else if constexpr (KIND == 0) wait_vm<NP>();
            else if constexpr (KIND == 1) { if (post) wait_vm<NP + E>(); else wait_vm<NP>(); }
            else if constexpr (KIND == 2) { if (more) wait_vm<NP>(); else wait_vm<0>(); }
            else { if (more) wait_vm<NP + R>(); else wait_vm<0>(); }
            SEG_E(0);
            sync_fetch();
            mma(I0{}, I0{});
            mma(I0{}, I1{});
            sync_mma();
            if constexpr (KINDX == 4) {
                // every wave has passed this phase's wait: the previous tile's stores have all retired
                if (pub) sc.publish(tm_prev, lane);
            }
            read_A(1, buf);
            if (KIND < 2 || more) { ld_advance(std::integral_constant<int, KIND == 2 ? 1 : 0>{}); issue_A(0, 0, GA); issue_W(0, 0, GB); issue_W(1, 0, GB); }
            SEG_B(0);
            if constexpr (KINDX == 4) { if (pub) wait_vm<NP + 1>(); else wait_vm<NP>(); }
            else if constexpr (KIND == 0) wait_vm<NP>();
            else if constexpr (KIND == 1) { if (post) wait_vm<NP + E>(); else wait_vm<NP>(); }
            else if constexpr (KIND == 2) { if (more) wait_vm<NP>(); else wait_vm<0>(); }
            else { if (more) wait_vm<NP + R>(); else wait_vm<0>(); }
            SEG_E(0);
            sync_fetch();
            mma(I1{}, I1{});
            mma(I1{}, I0{});
            sync_mma();
            ++g;
        };
        if constexpr (PH2) {
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
            using K3 = std::integral_constant<int, 3>;
            ph2_ktile(K1{});
            HG_TR(r == 2);
            if constexpr (PUB) {
                ph2_ktile(std::integral_constant<int, 4>{});
                ph2_ktile(std::integral_constant<int, 5>{});
            }
            for (int kt = PUB ? 3 : 1; kt < klen - 2; ++kt) {
                ph2_ktile(K0{});
                HG_TR((r == 1 && kt >= nk - 5) || (r == 2 && kt <= 4));
            }
            ph2_ktile(K2{});
            HG_TR(r == 1);
            ph2_ktile(K3{});
            HG_TR(r == 1);
        } else {
        for (int kt = 0; kt < klen; ++kt, ++g) {
            const int buf = (g & 1) * STAGE;
            const bool more = g + 2 < S;          // a K-tile two positions ahead exists
            const bool post = post_ok;             // epilogue stores of the previous tile may still be pending
            const bool xl = (XPRE || ROLL) && kt == klen - 1;  // residual rows are fetched during the last K-tile
            // ---------------- P1: fetch A0(t), W0(t); refill A1(t+1); then quadrant (0,0)
            read_A(0, buf);
            read_W(I0{}, buf);
            if constexpr (LNC) {
                // (mean, rstd) of this tile's rows -> LDS, one small DMA per wave (BM/8 rows x 8 B), a K-tile
                // ahead of the last one: by the last P4's counted wait it is more than N1 operations old, and that
                // phase's barriers publish it to every wave before the epilogue (the waits run one operation
                // stricter until it has retired)
                if (kt == klen - 2 && lane < BM / 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + MR_OFF + wave * BM), 16, lane * 16,
                                                             (m0 + wave * (BM / 8)) * 8, 0, 0);
            }
            if (g + 1 < S) issue_A(1, 0, GA);     // A1 of position g+1 (ld state already at g+1)
            if constexpr (ROLL) {
                if (xl) {      // first ROLL_W residual chunks (+ the first row group's centre) of this tile
#pragma unroll
                    for (int c = 0; c < ROLL_W; ++c) xw[c] = chunk_load(c);
                    if constexpr (RLN) muw[0] = p.mu[chunk_row(0)];
                }
            }
            if constexpr (XPRE) {
                if (xl) {
#pragma unroll
                    for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                        for (int f = 0; f < MF; ++f) {
                            int m = m0 + ha * (BM / 2) + wm * MF * 16 + f * 16 + (lane & 15);
                            m = m < p.M ? m : p.M - 1;
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
            SEG_B(0);
            if (!more) wait_vm<0>();              // -> W1(t) landed (read in P2)
            else if (xl) wait_vm<N2 + R>();
            else if (post && kt <= 1) wait_vm<N2 + E>();
            else wait_vm<N2>();
            SEG_E(0);
            sync_fetch();
            mma(I0{}, I0{});
            sync_mma();
            // ---------------- P2: fetch W1(t); slot A0(t) is free -> A0(t+2); quadrant (0,1)
            read_W(I1{}, buf);
            if (more) { ld_advance(std::integral_constant<int, 2>{}); issue_A(0, 0, GA); }
            SEG_B(0);
            if (!more) wait_vm<0>();              // -> A1(t) landed (read in P3)
            else if (xl) wait_vm<N2 + R>();
            else if (post && kt == 0) wait_vm<N2 + E>();
            else wait_vm<N2>();
            SEG_E(0);
            sync_fetch();
            mma(I0{}, I1{});
            sync_mma();
            // ---------------- P3: fetch A1(t); slot W0(t) free -> W0(t+2); quadrant (1,1)
            read_A(1, buf);
            if (more) issue_W(0, 0, GB);
            sync_fetch();
            mma(I1{}, I1{});
            sync_mma();
            // ---------------- P4: slot W1(t) free -> W1(t+2); quadrant (1,0)
            if (more) issue_W(1, 0, GB);
            SEG_B(0);
            if (!more) wait_vm<0>();              // -> A0(t+1), W0(t+1) landed (read in the next P1)
            else if (xl) wait_vm<N1 + R>();
            else if (post && kt == 0) wait_vm<N1 + E>();
            else wait_vm<N1>();
            SEG_E(0);
            sync_fetch();
            mma(I1{}, I0{});
            sync_mma();
                            }
        }
        // ---------------- epilogue of tile r (the ring keeps prefetching the next tile meanwhile)
        SEG_B(7);
        if (xmode & 4) {   // timing experiment: no epilogue
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int f = 0; f < MF; ++f)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) asm volatile("" ::"v"(acc[a][b][f][g2]));
            continue;
        }
        constexpr bool F16OUT = (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16 || LNC);
        if constexpr (F16OUT) {
            // fp16 outputs: a lane holds 4 consecutive columns (8 B) of one row.  v_permlane16_swap pairs the
            // accumulator tiles f, f+1 (same columns, rows 16 apart) so that even 16-lane groups end up with 8
            // consecutive columns of tile f's row and odd groups with 8 of tile f+1's row: 16-byte stores,
            // half the store instructions (the tail is store-issue bound).
            // tiles that lie entirely inside M (all of them at M = 197 * 256) skip the per-store row masks
            auto f16_epilogue = [&](auto INTERIOR_T) {
            constexpr bool INTERIOR = decltype(INTERIOR_T)::value;
            const int q = lane >> 4, r16 = lane & 15;
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            f32x4 gk = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_LN_BIAS_QGELU_F16) gk = quick_gelu_consts();
            (void)gk;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            // Column map (see voffW): acc[ha][hb][f][g2] of lane (r16, q) = row ha*BM/2 + wm*MF*16 + f*16 + r16, columns
            // wn*64 + 16q + 4(2hb + g2) + 0..3: the lane's four blocks are 16 consecutive columns = 32 bytes of fp16,
            // a row's 64 columns sit in its four q lanes.  A store instruction that touches 32 partial lines holds the
            // CU's store path for 72 cycles, one that writes 8 whole lines for 17 (tools/ubench/store_path.hip), so rows
            // r16 and r16 ^ 8 trade halves through a row_ror:8 DPP move: lanes r16 < 8 keep columns +0..7 and receive
            // +0..7 of row r16 + 8, lanes r16 >= 8 receive +8..15 of row r16 - 8 and keep their own +8..15; the first
            // store then writes rows 0..7 of the 16-row block and the second rows 8..15, eight lanes (128 B) per row.
            // Bias (and folded-weight column sums) of the lane's 16 columns are read once per tile.
            f32x4 bv[4], cv[4];
            const int nq = n0 + wn * 64 + 16 * q;
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                bv[b4] = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (nq + 4 * b4) * 4);
                if constexpr (LNC) cv[b4] = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + p.N * 4 + (nq + 4 * b4) * 4);
                else cv[b4] = bv[b4];
            }
            auto cvt2 = [](float a, float b) {      // RNE, one v_cvt_pk_f16_f32
                const half2v h = __builtin_convertvector(f32x2{a, b}, half2v);
                return __builtin_bit_cast(unsigned, h);
            };
            const bool low = r16 < 8;
            // byte offset of this lane's 16-byte piece inside a row: columns wn*64 + 8 * (2q + (r16 >> 3))
            half_t* colp = outp + n0 + wn * 64 + 8 * (2 * q + (r16 >> 3));
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int f = 0; f < MF; ++f) {
                    const int mb = m0 + ha * (BM / 2) + wm * MF * 16 + f * 16;      // first row of the 16-row block
                    f32x2 mr = {0.f, 1.f};                                           // (mean, rstd) of row mb + r16
                    if constexpr (LNC) mr = *reinterpret_cast<const f32x2*>(smem + MR_OFF + (mb + r16 - m0) * 8);
                    unsigned d[8];
#pragma unroll
                    for (int b4 = 0; b4 < 4; ++b4) {
                        f32x4 v;
                        if constexpr (LNC) v = (acc[ha][b4 >> 1][f][b4 & 1] - cv[b4] * mr[0]) * mr[1] + bv[b4];   // rstd * (acc - mean * cs) + bias'
                        else v = acc[ha][b4 >> 1][f][b4 & 1] + bv[b4];
                        if constexpr (EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_LN_BIAS_QGELU_F16) v = quick_gelu4(v, gk);
                        if constexpr (EPI == EPI_BIAS_RELU_F16) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        d[2 * b4] = cvt2(v[0], v[1]);
                        d[2 * b4 + 1] = cvt2(v[2], v[3]);
                    }
                    u32x4 st0, st1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned send = low ? d[4 + j] : d[j];
                        const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xF, 0xF, false);   // row_ror:8
                        st0[j] = low ? d[j] : recv;
                        st1[j] = low ? recv : d[4 + j];
                    }
                    const int row0 = mb + (r16 & 7);
                    if (INTERIOR || row0 < p.M) *reinterpret_cast<u32x4*>(colp + (size_t)row0 * p.ldc) = st0;
                    if (INTERIOR || row0 + 8 < p.M) *reinterpret_cast<u32x4*>(colp + (size_t)(row0 + 8) * p.ldc) = st1;
                }
            };
            if (m0 + BM <= p.M) f16_epilogue(std::true_type{});
            else f16_epilogue(std::false_type{});
        } else if constexpr (ROLL) {
            // residual epilogue through the rolling window: chunk c is consumed, stored, and its window slot is
            // refilled with chunk c + ROLL_W (the sched_barrier keeps the compiler from hoisting the refills)
            const int q = lane >> 4;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int rg = 0; rg < 2 * MF; ++rg) {
                const int ha = rg / MF, f = rg % MF;
                const int m = m0 + ha * (BM / 2) + wm * MF * 16 + f * 16 + (lane & 15);
                f32x4 v[2][2];
                float mu_r = 0.f;
                if constexpr (RLN) {
                    mu_r = muw[rg & 1];
                    if (rg + 1 < 2 * MF) muw[(rg + 1) & 1] = p.mu[chunk_row(rg + 1)];
                }
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int c = (rg * 2 + hb) * 2 + g2;
                        const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * q;
                        f32x4 a = acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4);
                        if constexpr (EPI == EPI_SCALE_RESID_F32) a *= *reinterpret_cast<const f32x4*>(p.pos + n);
                        v[hb][g2] = xw[c % ROLL_W] + a;
                        if (m < p.M)
                            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n) = v[hb][g2];
                        if (c + ROLL_W < 8 * MF) xw[c % ROLL_W] = chunk_load(c + ROLL_W);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                if constexpr (RLN) {
                    // per row and per wave column group (64 columns): (sum, sum of squared deviations from the group
                    // mean); fp16 copy centred on the row's previous mean
                    float sum = 0.f;
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                        for (int g2 = 0; g2 < 2; ++g2) sum += (v[hb][g2][0] + v[hb][g2][1]) + (v[hb][g2][2] + v[hb][g2][3]);
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
                    if (q == 0 && m < p.M) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        *reinterpret_cast<f32x2*>(p.stats + ((size_t)m * p.stats_ld + (n0 / 256) * 4 + wn) * 2) = f32x2{sum, m2};
                    }
                    // fp16 copy: the column blocks g2 = 0, 1 of this row are paired through v_permlane16_swap
                    // (even 16-lane groups end up with 8 consecutive columns of block 0, odd groups of block 1)
#pragma unroll
                    for (int hb = 0; hb < 2; ++hb) {
                        half4 h0, h1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h0[e] = (half_t)(v[hb][0][e] - mu_r);
                            h1[e] = (half_t)(v[hb][1][e] - mu_r);
                        }
                        const u32x2 ux = __builtin_bit_cast(u32x2, h0), uy = __builtin_bit_cast(u32x2, h1);
                        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        const int nc = n0 + hb * 128 + wn * 32 + ((q & 1) ? 16 : 0) + 4 * (q & ~1);
                        if (m < p.M) *reinterpret_cast<u32x4*>(p.out2 + (size_t)m * p.ld2 + nc) = o;
                    }
                }
            }
        } else {
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
            for (int f = 0; f < MF; ++f) {
                const int m = m0 + ha * (BM / 2) + wm * MF * 16 + f * 16 + (lane & 15);
#pragma unroll
                for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int n = n0 + hb * 128 + wn * 32 + g2 * 16 + 4 * (lane >> 4);
                        f32x4 v = acc[ha][hb][f][g2] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + n * 4);
                        if constexpr (RESID) {
                            if (m < p.M) {
                                if constexpr (EPI == EPI_SCALE_RESID_F32) v *= *reinterpret_cast<const f32x4*>(p.pos + n);
                                f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n);
                                if constexpr (XPRE) *dst = xres[ha][hb][f][g2] + v;
                                else *dst = *dst + v;
                            }
                        } else {
                            epilogue_ring<EPI>(p, m, n, v);
                        }
                    }
            }
        }
        SEG_E(7);
        HG_TR(r == 1);
        if constexpr (PUB) tm_prev = tm;
    }
    if constexpr (PUB) {      // the last tile: every wave drains its stores and meets the others before wave 0 publishes
        wait_vm<0>();
        barrier_raw();
    }
#ifdef HG_TRACE
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 8 + wave) * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = ttr[k];
    }
#endif
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
    if constexpr (PUB) {      // (waves 0-3 are here once waves 4-7 have passed the barrier behind their drain)
        if (wave == 0) sc.publish(tm_prev, lane);
    }
#endif
}

}  // namespace hg
