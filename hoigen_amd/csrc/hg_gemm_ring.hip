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

#include "hg_gemm_ring_body.h"

namespace hg {

// the stand-alone kernel: the body over the XCD-chunked, n-group-major tile list
template <int MF, int EPI, bool PH2>
__global__ __launch_bounds__(512, 2) void gemm_ring(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                    const unsigned a_bytes, const int mode, const int gsz) {
#if defined(__HIP_DEVICE_COMPILE__)
    const RingTileList sc(n_tiles, tiles_n, gsz);
    if (sc.n_items() <= 0) return;
    gemm_ring_body<MF, EPI, PH2>(p, a_bytes, mode, sc);
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
