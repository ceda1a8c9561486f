// Persistent fp16 MFMA GEMM with an LDS ring of half-tiles (gfx950), the workhorse for the large
// transformer GEMMs (M = B*197 = 50 432 rows at batch 256).
//
//   D[m][n] = sum_k A[m][k] * W[n][k]      A: activations [M,K] fp16, W: nn.Linear weight [N,K] fp16
//
// Geometry: tile (64*MF) x 256 x 64 with MF = 4 (256x256) or MF = 2 (128x256); 512 threads = 8 waves as
// 2(M) x 4(N); a wave owns (32*MF... rows from each A half) x (32 columns from each W half), i.e. four
// quadrants acc[ha][hb] of (16*MF) x 32 -> 32*MF accumulator VGPRs.  One workgroup per CU, persistent:
// it walks its list of output tiles and treats all their K-tiles as ONE stream, so the operand
// prefetch of the next output tile is already in flight while the current tile's epilogue runs.
//
// LDS ring: 2 K-tiles x {A0, A1, W0, W1} half-tile slots (A half = 32*MF rows, W half = 128 rows, 128 B
// per row, XOR-swizzled via the DMA source address).  A slot is refilled (buffer_load ... lds, 16 B
// per lane, no VGPR round trip) as soon as its fragments are in registers:
//
//   phase  fetch segment: LDS reads -> regs   DMA refill issued   vmcnt before its barrier      MFMA segment
//   P1(t)  A0(t) W0(t)                         A1(t+1)             (3GA+2GB) -> W1(t) landed     quadrant (A0,W0)
//   P2(t)  W1(t)                               A0(t+2)             (3GA+2GB) -> A1(t) landed     quadrant (A0,W1)
//   P3(t)  A1(t)                               W0(t+2)             -                             quadrant (A1,W1)
//   P4(t)  -                                   W1(t+2)             (2GA+3GB) -> A0,W0(t+1)       quadrant (A1,W0)
//
// (GA/GB = DMA instructions per wave per A/W half-tile.)  Every phase is [fetch segment] barrier [16 MFMAs]
// barrier, and waves 4-7 run one barrier interval behind waves 0-3: on each SIMD one wave is always in a
// fetch segment (LDS reads, DMA issue, counted waits) while its partner feeds the matrix pipe.  Up to five
// half-tiles (80 KiB at MF = 4) are in flight across the barriers; vmcnt is never drained in the loop.
// Epilogue stores (and the residual prefetch loads) count in vmcnt too, so the waits that follow them allow
// for E (R) more.
//
// PH2 (the default for 256x256 tiles): the same ring as TWO phases per K-tile - PA fetches A0 W0 W1, refills
// A1(t+1) and runs quadrants (A0,W0) (A0,W1); PB fetches A1, refills A0 W0 W1(t+2) and runs (A1,W1) (A1,W0); 32 MFMAs
// per segment, half the barriers, 2GA+2GB DMA instructions in flight at both waits.  Its K-tiles are instantiated by
// position in the tile (first / middle / second to last / last): the middle of the loop has no run-time conditions,
// and the load stream wraps to the next tile at a fixed K-tile.
//
// Epilogues: bias (+QuickGELU / ReLU) -> fp16 with v_permlane16_swap-paired 16-byte stores; LayerNorm-folded
// variants (EPI_LN_*: per-row (mean, rstd) staged in LDS by a small DMA, rstd * (acc - mean * cs) + b'); fp32
// residual with a full prefetch (MF = 2) or a rolling register window (MF = 4), optionally emitting the centred
// fp16 copy and row statistics (EPI_RESID_LN_F32).  Tiles that lie inside M skip the per-store row masks.
//
// The MFMA is issued with W rows as the A operand and activation rows as the B operand, so a lane
// holds 4 consecutive output columns of one row: 8-byte (fp16) / 16-byte (fp32) epilogue accesses.
#include <stdio.h>
#include <stdlib.h>

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

template <int MF, int EPI, bool PH2>
__global__ __launch_bounds__(512, 2) void gemm_ring(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                    const unsigned a_bytes, const int mode, const int gsz) {
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

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    // ---- this workgroup's tile list: round r -> tile id (XCD-contiguous chunks of 32 tiles)
    // Tile order (L2 locality): the list is n-group-major (groups of `gsz` column tiles whose W slices fit one
    // XCD's L2 together), m-tile next, column tile inside the group fastest.  XCD x (= blockIdx % 8 under
    // round-robin placement; speed only) owns the contiguous list range [x*T8, (x+1)*T8) and its CUs walk it
    // 'cpx' items per round, so an XCD keeps re-using the same W slices while streaming A panels.
    const int G = gridDim.x, bid = blockIdx.x;
    const bool xcd_ok = (G & 7) == 0;
    const int cpx = xcd_ok ? (G >> 3) : G;                         // workgroups per XCD
    const int T8 = xcd_ok ? (n_tiles + 7) / 8 : n_tiles;            // list items per XCD
    const int xbase = xcd_ok ? (bid & 7) * T8 : 0;
    const int xend = (xbase + T8 < n_tiles) ? xbase + T8 : n_tiles;
    const int slot = xbase + (xcd_ok ? (bid >> 3) : bid);          // first list item of this workgroup
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
    };        // tiles slot, slot+G, ...
    const int n_items = my_tiles;
    // item e of this workgroup: its tile (every item sweeps all of K)
    auto item_get = [&](int e, int& tm, int& tn, int& kb, int& ke) {
        kb = 0;
        ke = nk;
        tile_of(slot + e * cpx, tm, tn);
    };
    if (n_items <= 0) return;
    // De-synchronised epilogues: all tiles take the same time, so every CU would store (and, for the residual
    // epilogue, load) its output tile at the same moment - HBM idles during the K loops and saturates during
    // the epilogues.  Workgroups that own one tile fewer than the fullest ones have a tile time of slack; they
    // spend a pseudo-random fraction of it BEFORE their first tile instead of after their last.
    {
        const int dunit = mode >> 8;                               // estimated cycles per K-tile, 0 = off
        const int max_tiles = (T8 + cpx - 1) / cpx;
        if (dunit > 0 && my_tiles < max_tiles) {
            const unsigned h = ((unsigned)bid * 2654435761u) >> 24;   // 0..255
            const long long d = ((long long)(max_tiles - my_tiles) * nk * dunit * h) >> 8;
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
        auto ph2_ktile = [&](auto KIND_T) {
            constexpr int KIND = decltype(KIND_T)::value;
            const int buf = (g & 1) * STAGE;
            const bool post = post_ok;
            const bool more = KIND < 2 || r + 1 < n_items;       // a K-tile two positions ahead exists
            (void)post; (void)more;
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
            if constexpr (KIND == 0) wait_vm<NP>();
            else if constexpr (KIND == 1) { if (post) wait_vm<NP + E>(); else wait_vm<NP>(); }
            else if constexpr (KIND == 2) { if (more) wait_vm<NP>(); else wait_vm<0>(); }
            else { if (more) wait_vm<NP + R>(); else wait_vm<0>(); }
            SEG_E(0);
            sync_fetch();
            mma(I0{}, I0{});
            mma(I0{}, I1{});
            sync_mma();
            read_A(1, buf);
            if (KIND < 2 || more) { ld_advance(std::integral_constant<int, KIND == 2 ? 1 : 0>{}); issue_A(0, 0, GA); issue_W(0, 0, GB); issue_W(1, 0, GB); }
            SEG_B(0);
            if constexpr (KIND == 0) wait_vm<NP>();
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
            for (int kt = 1; kt < klen - 2; ++kt) {
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
#endif
}

template <int MF, int EPI, bool PH2 = false>
static hipError_t launch_ring_t(const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    if (!a.ld2) a.ld2 = a.ldc;
    constexpr int BM = 64 * MF;
    constexpr int RING = 2 * (2 * MF * 4096 + 2 * 16384);
    constexpr bool LNC = (EPI == EPI_LN_BIAS_F16 || EPI == EPI_LN_BIAS_QGELU_F16);
    const int LDS = RING + a.N * 4 * (LNC ? 2 : 1) + (LNC ? BM * 8 : 0);   // ring + bias[N] (+ cs[N] + (mean, rstd)[BM])
    if (LDS > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set_d[HG_MAX_DEVICES] = {};      // function attributes and CU counts are per device
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    bool& attr_set = attr_set_d[dev_i];
    int& n_cu = n_cu_d[dev_i];
    if (!attr_set) {
        n_cu = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring<MF, EPI, PH2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / 256;
    const int n_tiles = tiles_m * tiles_n;
    int grid = n_tiles < n_cu ? n_tiles : n_cu;
    // Even rounds: the tiles take `rounds` passes over the CUs whatever the grid, so use only as many workgroups as fill every
    // round (a multiple of 8: the XCD-chunked tile order needs it).  c_fc at batch 256: 2364 tiles = 10 rounds on 240
    // workgroups instead of 9.23 on 256 - the same number of rounds with fewer CUs contending for L2 / HBM: 236-239 -> 230 us
    // (round 3, one box; QKV's 1773 tiles stay on 256).  (HG_RING_GRID overrides in -DHG_EXPERIMENTS builds.)
    if (MF == 4 && n_tiles > n_cu) {
        const int rounds = (n_tiles + n_cu - 1) / n_cu;
        const int g8 = (((n_tiles + rounds - 1) / rounds) + 7) & ~7;
        if (g8 < grid) grid = g8;
    }
#ifdef HG_EXPERIMENTS
    if (const char* e = getenv("HG_RING_GRID")) { const int v = atoi(e); if (v >= 8 && v <= n_cu && v <= n_tiles) grid = v; }
#endif
    const size_t a_bytes = (size_t)tiles_m * BM * a.lda * 2;      // A is allocated with rows padded to 256
    // start stagger: estimated cycles per K-tile (a deliberate under-estimate).  The timing-experiment switches HG_RING_MODE,
    // HG_RING_DELAY (0 = no stagger) and HG_RING_GSZ are read in -DHG_EXPERIMENTS builds only
#ifdef HG_EXPERIMENTS
    static const int mode = []() {
        const char* e = getenv("HG_RING_MODE");
        const char* d = getenv("HG_RING_DELAY");
        return (e ? atoi(e) & 0xFF : 0) | ((d ? atoi(d) : (MF == 4 ? 3000 : 1800)) << 8);
    }();
    static const int gsz_env = []() { const char* e = getenv("HG_RING_GSZ"); return e ? atoi(e) : 0; }();
#else
    constexpr int mode = (MF == 4 ? 3000 : 1800) << 8;
    constexpr int gsz_env = 0;
#endif
    // column tiles per L2 group: W slices of one group (gsz * 256 rows * K * 2 B) should fit ~1.5 MiB, but
    // never fewer than 3: an A panel that is not shared by neighbouring column tiles is re-read from HBM once
    // per column tile (c_proj, K = 3072: 310 MB of activations x 3)
    int gsz = gsz_env > 0 ? gsz_env : (int)((1536 * 1024) / ((size_t)512 * a.K));
    if (gsz < 3) gsz = 3;
    if (gsz > tiles_n) gsz = tiles_n;
    if (gsz_env <= 0) {                                  // equal groups: 9 column tiles -> 3 + 3 + 3, not 4 + 4 + 1
        const int ngroups = (tiles_n + gsz - 1) / gsz;
        gsz = (tiles_n + ngroups - 1) / ngroups;
    }
#ifdef HG_TRACE
    if (getenv("HG_TRACE")) {
        const size_t n = (size_t)grid * 8 * 16;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipMemsetAsync(d, 0, n * 8, s);
        GemmArgs b = a;
        b.dbg = d;
        hipLaunchKernelGGL((gemm_ring<MF, EPI, PH2>), dim3(grid), dim3(512), LDS, s, b, tiles_n, n_tiles, (unsigned)a_bytes, mode, gsz);
        hipStreamSynchronize(s);
        unsigned long long* h = (unsigned long long*)malloc(n * 8);
        hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
        for (int blk : {0, 9, 130}) {
            if (blk >= grid) continue;
            const unsigned long long t0 = h[(size_t)(blk * 8) * 16];
            for (int w : {0, 4}) {
                fprintf(stderr, "[trace] ring<%d,%d> N=%d K=%d block %d wave %d:", MF, EPI, a.N, a.K, blk, w);
                for (int k = 0; k < 16; ++k) {
                    const unsigned long long t = h[(size_t)(blk * 8 + w) * 16 + k];
                    if (t) fprintf(stderr, " %lld", (long long)(t - t0));
                }
                fprintf(stderr, "\n");
            }
        }
        free(h);
        hipFree(d);
        return hipGetLastError();
    }
#endif
#ifdef HG_STAMPS
    if (getenv("HG_STAMPS")) {
        const size_t n = (size_t)grid * 8 * 16;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
        hipMemsetAsync(d, 0, n * 8, s);
        GemmArgs b = a;
        b.dbg = d;
        hipLaunchKernelGGL((gemm_ring<MF, EPI, PH2>), dim3(grid), dim3(512), LDS, s, b, tiles_n, n_tiles, (unsigned)a_bytes, mode, gsz);
        hipStreamSynchronize(s);
        unsigned long long* h = (unsigned long long*)malloc(n * 8);
        hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
        static const char* names[8] = {"vmcnt", "lgkmcnt", "fetch-barrier", "MFMA", "mfma-barrier", "DMA-issue", "ds_read-issue", "epilogue"};
        for (int w = 0; w < 8; w += 4) {
            double acc[10] = {0};
            for (int blk = 0; blk < grid; ++blk)
                for (int k = 0; k < 10; ++k) acc[k] += (double)h[(size_t)(blk * 8 + w) * 16 + k];
            const double kts = acc[9] > 0 ? acc[9] : 1;
            fprintf(stderr, "[stamps] ring<%d,%d> N=%d K=%d wave %d: loop %.0f cycles/K-tile;", MF, EPI, a.N, a.K, w, acc[8] / kts);
            for (int k = 0; k < 8; ++k)
                if ((HG_STAMP_MASK >> k) & 1) fprintf(stderr, " %s %.0f", names[k], acc[k] / kts);
            fprintf(stderr, "\n");
        }
        free(h);
        hipFree(d);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((gemm_ring<MF, EPI, PH2>), dim3(grid), dim3(512), LDS, s, a, tiles_n, n_tiles, (unsigned)a_bytes, mode, gsz);
    return hipGetLastError();
}

// Ring kernel eligibility: N % 256 == 0, K >= 256 (>= 4 K-tiles so at most one epilogue's stores are in
// flight inside the vmcnt window), everything addressable with 32-bit byte offsets.
bool gemm_ring_ok(const GemmArgs& a) {
    if (a.N % 256 || a.K % 64 || a.K < 256 || a.M < 512 || a.N > 8192) return false;
    const size_t Mp = (size_t)((a.M + 255) / 256) * 256;
    if (Mp * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;
    return true;
}

bool gemm_ln_ok(int epi, const GemmArgs& a) {
    if (!gemm_ring_ok(a)) return false;
    if (epi == EPI_RESID_LN_F32) return gemm_ring2_ok(a) && a.out2 && a.stats && a.mu && a.stats_ld == 4 * (a.N / 256);
    return a.cs && a.mr && a.N <= 3584;      // 128 KiB ring + 2 * N * 4 + 2 KiB of LDS; mr readable for padded rows
}

// 256x256 tiles: two phases per K-tile, except with the residual epilogues (their rolling window does not fit beside the
// two-phase loop's fragment registers: those instantiations would spill and are never built)
template <int E>
static hipError_t launch_big(const GemmArgs& a, hipStream_t s, bool ph2) {
    constexpr bool resid = (E == EPI_BIAS_RESID_F32 || E == EPI_SCALE_RESID_F32 || E == EPI_RESID_LN_F32);
    if constexpr (resid) return launch_ring_t<4, E, false>(a, s);
    else return ph2 ? launch_ring_t<4, E, true>(a, s) : launch_ring_t<4, E, false>(a, s);
}

hipError_t launch_gemm_ring(int epi, const GemmArgs& a, hipStream_t s) {
    const bool lnc = (epi == EPI_LN_BIAS_F16 || epi == EPI_LN_BIAS_QGELU_F16);   // this kernel only
    const bool resid = (epi == EPI_BIAS_RESID_F32 || epi == EPI_SCALE_RESID_F32 || epi == EPI_RESID_LN_F32);
    // 256x256 tiles when they fill the chip evenly enough, else 128x256 (N = 768 GEMMs: 591 vs 1182 tiles).
    // The 128x256 kernel needs 50 % more global->LDS traffic per FLOP and runs at ~0.8 of a 256x256 tile's time
    // per (half-size) tile, so with the residual epilogues (rolling window at 256 rows) the big tile wins as soon
    // as rounds(256) <= 0.8 * rounds(128): N = 768, M = 50432: 3 vs 5 * 0.8.
    const int t256 = ((a.M + 255) / 256) * (a.N / 256), t128 = ((a.M + 127) / 128) * (a.N / 256);
    const int rounds = (t256 + 255) / 256, rounds128 = (t128 + 255) / 256;
#ifdef HG_EXPERIMENTS
    static const int force_big = []() { const char* e = getenv("HG_RING_BIG"); return e ? atoi(e) : 0; }();
    static const bool ph2 = []() { const char* e = getenv("HG_RING_PH2"); return e ? atoi(e) != 0 : true; }();
#else
    constexpr int force_big = 0;
    constexpr bool ph2 = true;
#endif
    bool big = t256 >= 256 && (double)t256 / (rounds * 256.0) >= 0.9;
    // ... or when its whole passes (even rounds: launch_ring_t uses only as many workgroups as fill them) still beat the
    // 128-row kernel's: text-tower QKV, M = 46 200, N = 1536: 5 passes of 256x256 against 8.5 of 128x256 at ~0.8 of the time
    // each (80 -> 73 us, round 3)
    if (!resid && !big && t256 >= 256 && (double)rounds <= 0.8 * (double)t128 / 256.0) big = true;
    // ... measured at N = 768, M = 50432: plain residual 272 vs 298 us (K = 3072).  With the LayerNorm extras
    // (fp16 copy + statistics: 387 MB per launch) the epilogue is an HBM burst of every CU at once, and three big
    // bursts overlap worse than five small ones (K = 768: 154 vs 127 us; K = 3072: 300 vs 303): keep 128 rows.
    if (resid && epi != EPI_RESID_LN_F32 && t256 >= 256 && (double)rounds <= 0.8 * rounds128 + 1e-9) big = true;
    if (force_big == 1) big = true;
    if (force_big == 2 || force_big == 3) big = false;
    if (epi == EPI_RESID_LN_F32 && (!big || a.hl || a.gamma)) return launch_gemm_ring2(epi, a, s);     // no 128-row variant in this kernel; the hi / lo stream lives in ring2's tile order
    if (!big && force_big != 3 && !lnc && gemm_ring2_ok(a)) return launch_gemm_ring2(epi, a, s);   // 128x256, two-phase
    // 256x256 tiles run two phases per K-tile (32 MFMAs per segment): -4..8 % vs four phases (HG_RING_PH2=0 in experiments builds)
    // ... except with the residual epilogues: their rolling window does not fit beside the two-phase loop's fragment
    // registers (9-21 spilled VGPRs, 4-10 % slower than four phases)
#define HG_RING(E)                                                         \
    case E:                                                                \
        return big ? launch_big<E>(a, s, ph2) : launch_ring_t<2, E, false>(a, s)
    switch (epi) {
        HG_RING(EPI_BIAS_F16);
        HG_RING(EPI_BIAS_QGELU_F16);
        HG_RING(EPI_BIAS_RELU_F16);
        HG_RING(EPI_BIAS_RESID_F32);
        HG_RING(EPI_BIAS_F32);
        HG_RING(EPI_PATCH_F32);
        HG_RING(EPI_BIAS_RELU_F32);
        HG_RING(EPI_SCALE_RESID_F32);
        HG_RING(EPI_LN_BIAS_F16);
        HG_RING(EPI_LN_BIAS_QGELU_F16);
        case EPI_RESID_LN_F32: return launch_ring_t<4, EPI_RESID_LN_F32, false>(a, s);
        default: return hipErrorInvalidValue;
    }
#undef HG_RING
}

}  // namespace hg
