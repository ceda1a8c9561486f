// Row-owner residual GEMM for gfx950: every workgroup owns WHOLE ROWS of the residual stream (N = 96 * 8 = 768 or
// 64 * 8 = 512 columns), so the activation panel is fetched from HBM exactly once, the LayerNorm statistics of the updated
// rows are finished inside the kernel (no partial-statistics buffer, no finalize launch), and the weight - the operand that
// is shared by every workgroup - streams from L2 as contiguous 1 KiB MFMA fragments of a layout packed once at load time.
//
//   x[m][:] += A[m][:] W^T + bias         (clipnet/model.py:185-188: the attention out-projection and the MLP c_proj)
//   x16[m][:] = fp16(x[m][:] - mu[m])     centred copy for the next LayerNorm-folded GEMM        (RLN)
//   mr[m] = (mean - mu[m], rstd), mu[m] = mean                                                   (RLN)
//
// Geometry: 512 threads = 8 waves, all along N: wave w owns columns [w * 16 * NCB, (w + 1) * 16 * NCB) of every row of the
// tile (NCB = 6 column blocks of 16 at N = 768), i.e. RB x NCB accumulator blocks of 16 x 16 (RB <= 7 row blocks: 168 VGPRs).
// A workgroup's rows (an even share of the 16-row blocks: 12 or 13 at batch 256 = 197 rows per CU) are walked as tiles of at
// most 7 row blocks; every tile sweeps K once.
//
// Operand paths (all buffer_load ... lds, counted vmcnt, one s_barrier per K-tile):
//   A (activations): shared ring of three K-tile stages of RB x 16 rows x 128 B (XOR-swizzled chunks as in hg_gemm_ring.hip);
//                    each wave issues two 1 KiB pieces per K-tile, two K-tiles ahead; all eight waves read every row.
//   W (weights):     wave-PRIVATE ring of one K-tile: 2 x NCB fragments of 1 KiB in lane order (ds_read_b128 at lane * 16, no
//                    swizzle, no sharing, no barrier); a slot is refilled with the next K-tile's fragment as soon as its
//                    MFMAs are issued.  Packed layout: Wp[k / 32][n / 16][lane][8] = W[16 (n/16) + lane % 16][32 (k/32) + 8 (lane / 16) + 0..7]
//                    (pack_w_frag_kernel), the A operand of v_mfma_f32_16x16x32_f16 as it stands.
// Bytes through the CU's load path per K-tile (RB = 7, N = 768): 14 KiB of A + 96 KiB of W for 2 x 7 x 48 MFMAs.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

namespace {
constexpr int ROWS_W_OFF = 0;                    // 8 waves x 2 * NCB KiB
constexpr int ROWS_A_STG = 16384;                // one A stage: up to 8 row blocks x 2 KiB (two pieces per wave)
constexpr int ROWS_MAXRB = 7;
}  // namespace

// One tile: rows [m0, m0 + 16 * RB) x all columns.  `first`: the operands of this tile's first K-tile (and the A stage of its
// second) have landed and vmcnt is drained of everything but stores (prologue / the epilogue in front of it saw to that).
// m0_next: row origin of the workgroup's next tile (the load stream runs on into it), or m0 again when there is none.
template <int RB, int NCB, bool RLN>
__device__ __forceinline__ void rows_tile(const GemmArgs& p, char* smem, const int m0, const int m0_next, const int rbn, int& stg,
                                          const __amdgpu_buffer_rsrc_t rsA, const __amdgpu_buffer_rsrc_t rsW,
                                          const int (&voffA)[2], const int lane, const int wave, const int mode) {
    // timing-experiment switches (HG_ROWS_MODE bits: 1 no epilogue, 2 no MFMA, 4 no W DMA, 8 no A DMA; wrong results) exist only
    // in a -DHG_EXPERIMENTS build
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int N = 8 * NCB * 16;
    constexpr int WSLOT = 2 * NCB * 1024;
    constexpr int A_OFF = 8 * WSLOT;
    constexpr int BIAS_OFF = A_OFF + 3 * ROWS_A_STG;
    constexpr int ST_OFF = BIAS_OFF + N * 4;
    const int nk = p.K >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    char* wring = smem + ROWS_W_OFF + wave * WSLOT;

    f32x4 acc[RB][NCB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // lane-constant LDS read offsets
    const int a_lane = r16 * 128;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + q) ^ ((lane >> 1) & 7)) << 4;

    auto issue_A = [&](int kt_ahead, int stage) {      // both pieces of this wave for K-tile kt_ahead of the stream
        const bool wrap = kt_ahead >= nk;
        const int kt = wrap ? kt_ahead - nk : kt_ahead;
        const int mo = wrap ? m0_next : m0;
        const int soff = (mo * p.lda + kt * 64) * 2;
        if (xmode & 8) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + A_OFF + stage * ROWS_A_STG + (wave + 8 * i) * 1024), 16,
                                                     voffA[i], soff, 0, 0);
    };
    auto issue_W = [&](int kt_ahead, const int slot) {     // fragment slot (ks, c) of K-tile kt_ahead (wraps to the next tile's K-tile 0)
        const int ks = slot / NCB, c = slot % NCB;
        const int kt = kt_ahead >= nk ? kt_ahead - nk : kt_ahead;
        const int soff = (((kt * 2 + ks) * (8 * NCB)) + wave * NCB + c) * 1024;
        if (xmode & 4) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(wring + slot * 1024), 16, lane * 16, soff, 0, 0);
    };
    auto read_W = [&](int slot) { return *reinterpret_cast<const half8*>(wring + slot * 1024 + lane * 16); };

    // One K-tile = two k-steps of 32; per step every W fragment of the wave meets all RB activation fragments (RB MFMAs per
    // fragment).  The activation fragments of the NEXT step are read behind the step's last MFMAs, each into the register the
    // MFMA just issued has read: no second fragment set, no read burst of all eight waves behind the barrier.
    //   step 0 | wait A(kt+1) landed, s_barrier: publishes stage kt+1, frees stage kt | issue A(kt+3) | step 1
    // VMEM operations of a K-tile in issue order: R_0 .. R_{NCB-1}  A_0 A_1  R_NCB .. R_{2NCB-1}  (R_i = refill of W slot i with the
    // next K-tile's fragment, right behind the slot's MFMAs).  The wait in front of the read of slot i counts what has been
    // issued since R_i of the previous K-tile: 2 NCB operations, 2 NCB + 1 for slot 0, 2 NCB - 2 for slot NCB (read before the
    // A pieces of this K-tile are issued, refilled after those of the previous one).
    // KIND 0: first K-tile of a tile (its operands have landed, vmcnt holds at most stores: no waits; reads its own A
    // fragments), 1: middle, 2: last (no fragment prefetch across the epilogue).
    half8 af[RB];
    auto read_af = [&](int stage, int ks, int rb) {
        af[rb] = *reinterpret_cast<const half8*>(smem + A_OFF + stage * ROWS_A_STG + a_lane + rb * 2048 + coff[ks]);
    };
    auto ktile = [&](const int kt, auto KIND_T) {
        constexpr int KIND = decltype(KIND_T)::value;
        constexpr bool FIRST = KIND == 0, LAST = KIND == 2;
        const int st_cur = stg;
        const int st_nxt = stg == 2 ? 0 : stg + 1;
        stg = st_nxt;
        if constexpr (FIRST) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) read_af(st_cur, 0, rb);
        }
        half8 wf;
        if constexpr (!FIRST) wait_vm<2 * NCB + 1>();
        else asm volatile("" ::: "memory");      // (same scheduling fences as the waits: keeps the fragment reads where they are)
        wf = read_W(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int slot = ks * NCB + c;
                half8 wn = wf;
                if (slot + 1 < 2 * NCB) {
                    if constexpr (FIRST) asm volatile("" ::: "memory");
                    else if (slot + 1 == NCB) wait_vm<2 * NCB - 2>();
                    else wait_vm<2 * NCB>();
                    wn = read_W(slot + 1);
                }
                const bool pre = c == NCB - 1 && (ks == 0 || !LAST);
                if (xmode & 2) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        asm volatile("" ::"v"(af[rb]), "v"(wf));
                        if (pre) read_af(ks == 0 ? st_cur : st_nxt, ks == 0 ? 1 : 0, rb);
                    }
                } else {
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[rb], acc[rb][c], 0, 0, 0);
                        if (pre) {      // (pinned: a read hoisted above its MFMA would need a second register for the fragment)
                            __builtin_amdgcn_sched_barrier(0);
                            read_af(ks == 0 ? st_cur : st_nxt, ks == 0 ? 1 : 0, rb);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                }
                asm volatile("" ::: "memory");
                // the slot's fragment is in registers (the MFMAs above consumed it): refill it with the next K-tile's
                issue_W(kt + 1, slot);
                wf = wn;
            }
            if (ks == 0) {
                // A of K-tile kt+1 was issued two K-tiles ago (4 NCB + 2 operations back); in the FIRST K-tile it landed before
                // the tile began
                if constexpr (!FIRST) wait_vm<4 * NCB + 2>();
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of stage kt are done
                barrier_raw();
                issue_A(kt + 3, st_cur);
            }
        }
    };

    ktile(0, std::integral_constant<int, 0>{});
    for (int kt = 1; kt < nk - 1; ++kt) ktile(kt, std::integral_constant<int, 1>{});
    ktile(nk - 1, std::integral_constant<int, 2>{});
    // operands of the next tile's first K-tile (and its next two A stages) are in flight: land them before the epilogue's own
    // loads and stores enter vmcnt, and publish them (the next tile starts with a FIRST K-tile)
    wait_vm<0>();
    barrier_raw();
    if (xmode & 1) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCB; ++c) asm volatile("" ::"v"(acc[rb][c]));
        return;
    }

    // ---------------- epilogue: this wave's 16 * NCB columns of every row of the tile
    float* xo = reinterpret_cast<float*>(p.out);
    const int ncol0 = wave * 16 * NCB + 4 * q;                 // + 16 c
    auto row_of = [&](int rb) {
        const int m = m0 + rb * 16 + r16;
        return m < p.M ? m : p.M - 1;
    };
    // One pass per row block: the chunk registers are refilled with the next row block's residual as soon as a chunk is done
    // (six 1 KiB loads per wave in flight throughout).  Statistics as shifted sums about the copy's centre c = mu[m] (the row's
    // previous mean): s1 = sum (v - c), s2 = sum (v - c)^2 - the differences are what the fp16 copy stores anyway, and with c
    // within a fraction of the row's spread of its mean, s2 / N - (s1 / N)^2 loses nothing to cancellation.
    f32x4 xr[NCB];
    float mu_c, mu_n = 0.f;
    {
        const int m = row_of(0);
#pragma unroll
        for (int c = 0; c < NCB; ++c) xr[c] = *reinterpret_cast<const f32x4*>(xo + (size_t)m * p.ldc + ncol0 + 16 * c);
        if constexpr (RLN) mu_n = p.mu[m];
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int m = m0 + rb * 16 + r16;
        const bool ok = m < p.M && rb < rbn;
        const int mn = rb + 1 < RB ? row_of(rb + 1) : 0;
        mu_c = mu_n;
        if constexpr (RLN) {
            if (rb + 1 < RB) mu_n = p.mu[mn];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const f32x4 v = xr[c] + (acc[rb][c] + *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (ncol0 + 16 * c) * 4));
            if (rb + 1 < RB) xr[c] = *reinterpret_cast<const f32x4*>(xo + (size_t)mn * p.ldc + ncol0 + 16 * c);
            if (ok) *reinterpret_cast<f32x4*>(xo + (size_t)m * p.ldc + ncol0 + 16 * c) = v;
            if constexpr (RLN) {
                const f32x4 d = v - mu_c;
                s1 += (d[0] + d[1]) + (d[2] + d[3]);
                s2 = fmaf(d[0], d[0], s2); s2 = fmaf(d[1], d[1], s2); s2 = fmaf(d[2], d[2], s2); s2 = fmaf(d[3], d[3], s2);
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)d[e];
                if (ok) *reinterpret_cast<half4*>(p.out2 + (size_t)m * p.ld2 + ncol0 + 16 * c) = h;
            }
        }
        if constexpr (RLN) {
            s1 = sum_rows(s1);                                  // over the four lane groups: this wave's 16 * NCB columns of the row
            s2 = sum_rows(s2);
            if (q == 0) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2*>(smem + ST_OFF + ((rb * 16 + r16) * 8 + wave) * 8) = f32x2{s1, s2};
            }
        }
    }
    if constexpr (RLN) {
        // the eight column groups of a row meet in LDS; the LayerNorm statistics go straight to the consumer's table - what
        // finalize_stats_kernel did in its own launch
        __builtin_amdgcn_s_waitcnt(0xC07F);
        barrier_raw();
        const int t = wave * 64 + lane;
        if (t < RB * 16) {
            const int m = m0 + t;
            if (m < p.M && (t >> 4) < rbn) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 g[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) g[w] = *reinterpret_cast<const f32x2*>(smem + ST_OFF + (t * 8 + w) * 8);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { s1 += g[w][0]; s2 += g[w][1]; }
                const float dm = s1 * (1.0f / N);                       // mean - centre
                const float var = fmaxf(s2 * (1.0f / N) - dm * dm, 0.f);
                const float c0 = p.mu[m];
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2*>(p.mr_out + 2 * (size_t)m) = f32x2{dm, 1.0f / sqrtf(var + 1e-5f)};
                const float mean = c0 + dm;
                if (p.muc_out) p.muc_out[m] = c0;
                p.mu_out[m] = mean;
            }
        }
    }
}

template <int NCB, bool RLN>
__global__ __launch_bounds__(512, 2) void gemm_rows(const GemmArgs p, const int nb_total, const unsigned a_bytes, const int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int N = 8 * NCB * 16;
    constexpr int WSLOT = 2 * NCB * 1024;
    constexpr int A_OFF = 8 * WSLOT;
    constexpr int BIAS_OFF = A_OFF + 3 * ROWS_A_STG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, bid = blockIdx.x;
    // this workgroup's 16-row blocks [b0, b1): an even share; tiles of at most ROWS_MAXRB blocks, sizes as equal as they come
    const int b0 = (int)((long long)bid * nb_total / G), b1 = (int)((long long)(bid + 1) * nb_total / G);
    const int nbw = b1 - b0;
    if (nbw <= 0) return;
    const int nt = (nbw + ROWS_MAXRB - 1) / ROWS_MAXRB;
    const int tb = nbw / nt, trem = nbw - tb * nt;           // tiles 0 .. trem-1 have tb + 1 blocks
    auto tile_blocks = [&](int i) { return tb + (i < trem ? 1 : 0); };
    auto tile_first = [&](int i) { return b0 + i * tb + (i < trem ? i : trem); };

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)((size_t)N * p.K * 2), 0x00020000);
    // A pieces of this wave: piece pc = wave + 8 i covers tile rows 8 pc .. 8 pc + 7; lane -> (row = l >> 3, chunk' = l & 7)
    int voffA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffA[i] = row * p.lda * 2 + c * 16;
    }
    for (int i = tid; i < N / 4; i += 512)
        *reinterpret_cast<f32x4*>(smem + BIAS_OFF + i * 16) =
            p.bias ? reinterpret_cast<const f32x4*>(p.bias)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- prologue: A of stream positions 0, 1 and 2, W of position 0
    {
        const int m0 = tile_first(0) * 16;
        const int nk = p.K >> 6;
#pragma unroll
        for (int pos = 0; pos < 3; ++pos) {
            const int soff = (m0 * p.lda + (pos < nk ? pos : 0) * 64) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + A_OFF + pos * ROWS_A_STG + (wave + 8 * i) * 1024), 16,
                                                         voffA[i], soff, 0, 0);
        }
#pragma unroll
        for (int slot = 0; slot < 2 * NCB; ++slot) {
            const int ks = slot / NCB, c = slot % NCB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + ROWS_W_OFF + wave * WSLOT + slot * 1024), 16, lane * 16,
                                                     ((ks * (8 * NCB)) + wave * NCB + c) * 1024, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        barrier_raw();
    }
    int stg = 0;
    for (int i = 0; i < nt; ++i) {
        const int rbn = tile_blocks(i);
        const int m0 = tile_first(i) * 16;
        const int m0n = i + 1 < nt ? tile_first(i + 1) * 16 : m0;
        // (a tile of rbn blocks runs the next larger instantiation; the rows beyond rbn are computed and never stored)
        if (rbn == 7) rows_tile<7, NCB, RLN>(p, smem, m0, m0n, rbn, stg, rsA, rsW, voffA, lane, wave, mode);
        else if (rbn == 6) rows_tile<6, NCB, RLN>(p, smem, m0, m0n, rbn, stg, rsA, rsW, voffA, lane, wave, mode);
        else if (rbn == 5) rows_tile<5, NCB, RLN>(p, smem, m0, m0n, rbn, stg, rsA, rsW, voffA, lane, wave, mode);
        else rows_tile<4, NCB, RLN>(p, smem, m0, m0n, rbn, stg, rsA, rsW, voffA, lane, wave, mode);
    }
#endif
}

// ---- weight packing (load time) ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_w_frag_kernel(const half_t* __restrict__ W, half_t* __restrict__ Wp, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece: (k32, cb, lane)
    const size_t total = (size_t)(K / 32) * (N / 16) * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const size_t f = i >> 6;
    const int cb = (int)(f % (size_t)(N / 16)), k32 = (int)(f / (size_t)(N / 16));
    const half8 v = *reinterpret_cast<const half8*>(W + (size_t)(16 * cb + (lane & 15)) * K + 32 * k32 + 8 * (lane >> 4));
    *reinterpret_cast<half8*>(Wp + i * 8) = v;
}
hipError_t launch_pack_w_frag(const half_t* W, half_t* Wp, int N, int K, hipStream_t s) {
    if (N % 16 || K % 32) return hipErrorInvalidValue;
    const size_t total = (size_t)(K / 32) * (N / 16) * 64;
    hipLaunchKernelGGL(pack_w_frag_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, Wp, N, K);
    return hipGetLastError();
}

bool gemm_rows_ok(int epi, const GemmArgs& a) {
    if (epi != EPI_RESID_LN_F32 && epi != EPI_BIAS_RESID_F32) return false;
    if (!a.Wp || (a.N != 768 && a.N != 512) || a.K % 64 || a.K < 256 || a.M < 64 * 16) return false;
    if (epi == EPI_RESID_LN_F32 && (!a.out2 || !a.mu || !a.mr_out || !a.mu_out)) return false;
    const size_t Mp = (size_t)((a.M + 255) / 256) * 256;
    if (Mp * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;
    return true;
}

template <int NCB, bool RLN>
static hipError_t launch_rows_t(const GemmArgs& a, hipStream_t s) {
    constexpr int N = 8 * NCB * 16;
    constexpr int LDS = 8 * 2 * NCB * 1024 + 3 * ROWS_A_STG + N * 4 + ROWS_MAXRB * 16 * 8 * 8;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows<NCB, RLN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int nb_total = (a.M + 15) / 16;
    int grid = nb_total / 4;                      // at least four row blocks per workgroup
    if (grid > n_cu_d[dev_i]) grid = n_cu_d[dev_i];
    if (grid < 1) grid = 1;
    const size_t a_bytes = (size_t)((a.M + 255) / 256) * 256 * a.lda * 2;      // A is allocated with rows padded to 256
    static const int mode = []() { const char* e = getenv("HG_ROWS_MODE"); return e ? atoi(e) : 0; }();      // experiments build only
    hipLaunchKernelGGL((gemm_rows<NCB, RLN>), dim3(grid), dim3(512), LDS, s, a, nb_total, (unsigned)a_bytes, mode);
    return hipGetLastError();
}

hipError_t launch_gemm_rows(int epi, const GemmArgs& a_in, hipStream_t s) {
    GemmArgs a = a_in;
    if (!a.ld2) a.ld2 = a.ldc;
    if (!gemm_rows_ok(epi, a)) return hipErrorInvalidValue;
    const bool rln = epi == EPI_RESID_LN_F32;
    if (a.N == 768) return rln ? launch_rows_t<6, true>(a, s) : launch_rows_t<6, false>(a, s);
    return rln ? launch_rows_t<4, true>(a, s) : launch_rows_t<4, false>(a, s);
}

}  // namespace hg
