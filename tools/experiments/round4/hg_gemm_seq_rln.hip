// Residual GEMMs of a vision-tower block on sequence tiles (gfx950): the attention out-projection and the MLP c_proj
// (clipnet/model.py:185-188) with the LayerNorm-emitting epilogue of hg_gemm_ring2.hip (EPI_RESID_LN_F32):
//
//   x[m][:] += A[m][:] W^T + bias;   x16[m][:] = fp16(x[m][:] - mu[m]);   stats[m][group] = (sum, sum of squared deviations)
//
// on the K loop of hg_seq_dev.h: a work item is (sequence, 384-column panel), i.e. a 208 x 384 tile whose rows are ONE
// sequence; 2 panels at N = 768, so 512 items at batch 256 = exactly two per CU, the two panels of a sequence side by side on
// one XCD (the activation panel - 1.2 MB for c_proj - is fetched from HBM once and shared through that XCD's L2: the 128 x 256
// tiles of gemm_ring2 fetch it 2.1 times).  The loop moves 7.4 KB through the CU's load path per MFLOP (ring2: 11.4), takes one
// barrier per K-tile and never drains a wave at a K-tile boundary.
//
// Epilogue, per wave 13 row blocks x 48 columns: the residual rows arrive through a rolling window of four row blocks (the
// fragment registers are dead by then), every row block is updated, stored, reduced to its (sum, M2) over the wave's 48 columns
// (16 column groups of 48 per row: finalize_stats) and re-emitted as the centred fp16 copy, two row blocks per 16-byte store.
// HL (GemmArgs::hl): the stream as centre + hi + lo like gemm_ring2 - hi IS the copy, lo = fp16((x - mu) - hi) - but here lo is
// row-major [M, N] like hi (this kernel family is the only reader): 0 fp32 in / out, 1 fp32 in, hi + lo out, 2 hi + lo in / out,
// 3 hi + lo in, fp32 out.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_seq_dev.h"

namespace hg {

namespace {
constexpr int GS_BIAS = SQ_END;                    // bias of the panel [384]
constexpr int GS_MU = GS_BIAS + 384 * 4;           // centre the copy is written with [208]
constexpr int GS_MUC = GS_MU + SQ_RB * 16 * 4;     // centre the hi / lo being read were written with [208]
constexpr int GS_LDS = GS_MUC + SQ_RB * 16 * 4;
static_assert(GS_LDS <= 160 * 1024, "LDS budget");
}  // namespace

template <int HL>
__global__ __launch_bounds__(512, 2) void gemm_seq_kernel(const GemmArgs p, const half_t* __restrict__ wp, const int n_seq, const int L,
                                                          const unsigned a_bytes, const int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    // timing-experiment switches (HG_GS_MODE bits: 2 no MFMA in the K loop, 4 no epilogue, 8 no operand DMA; wrong results) exist only
    // in a -DHG_EXPERIMENTS build
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int RB = SQ_RB, NCB = SQ_NCB;
    constexpr bool IN_HL = (HL == 2 || HL == 3), OUT_HL = (HL == 1 || HL == 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 6;
    const int PN = p.N / 384;

    // ---- this workgroup's items: XCD x (= blockIdx % 8 under round-robin placement; speed only) owns the sequences
    // [x * spx, (x + 1) * spx); sequence-major, panel fastest: the panels of a sequence run side by side
    const int G = gridDim.x, bid = blockIdx.x;
    const bool xcd_ok = (G & 7) == 0;
    const int cpx = xcd_ok ? (G >> 3) : G;
    const int idx = xcd_ok ? (bid >> 3) : bid;
    const int spx = xcd_ok ? ((n_seq + 7) >> 3) : n_seq;
    const int s0 = xcd_ok ? (bid & 7) * spx : 0;
    int ns = n_seq - s0;
    ns = ns < 0 ? 0 : (ns > spx ? spx : ns);
    const int nx = ns * PN;
    if (idx >= nx) return;
    auto decode = [&](int e, int& seq, int& pn) {
        const int s = e / PN;
        seq = s0 + s;
        pn = e - s * PN;
    };

#define SQ_A_PTR p.A
#define SQ_A_BYTES a_bytes
#define SQ_LDA p.lda
#define SQ_W_PTR wp
#define SQ_W_BYTES (unsigned)((size_t)p.N * p.K * 2)
#include "hg_seq_kloop.inc"

    // bias of the panel (1.5 KiB: waves 0, 1), mu (and muc) of the sequence's rows (208 x 4 B: waves 2, 3)
    auto issue_extras = [&](int seq, int pn) {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)(p.N * 4), 0x00020000);
        if (wave == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (HG_LDS void*)(smem + GS_BIAS), 16, lane * 16, pn * 384 * 4, 0, 0);
        else if (wave == 1) {
            if (lane < 32)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (HG_LDS void*)(smem + GS_BIAS + 1024), 16, lane * 16, pn * 384 * 4 + 1024, 0, 0);
        } else if (wave == 2) {
            const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.mu, 0, (unsigned)((size_t)n_seq * L * 4), 0x00020000);
            if (lane < RB * 16 * 4 / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + GS_MU), 16, lane * 16, seq * L * 4, 0, 0);
        } else if (wave == 3 && IN_HL) {
            const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.muc, 0, (unsigned)((size_t)n_seq * L * 4), 0x00020000);
            if (lane < RB * 16 * 4 / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (HG_LDS void*)(smem + GS_MUC), 16, lane * 16, seq * L * 4, 0, 0);
        }
    };

    int e = idx, seq, pn;
    decode(e, seq, pn);
    seq_prologue(seq * L, pn);

    for (;;) {
        const int e_n = e + cpx;
        const bool has_next = e_n < nx;
        int seq_n = seq, pn_n = pn;
        if (has_next) decode(e_n, seq_n, pn_n);

        issue_extras(seq, pn);
        f32x4 acc[RB][NCB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCB; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int sq_row0 = seq * L, sq_pn = pn, sq_row0_n = seq_n * L, sq_pn_n = pn_n;
#include "hg_seq_kloop_run.inc"
        }

        if (xmode & 4) {
            wait_vm<0>();
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int c = 0; c < NCB; ++c) asm volatile("" ::"v"(acc[rb][c]));
        } else {
            // ---------------- epilogue (opaque lane id: its lane constants must not be hoisted above the K loop, where every register is taken)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int q = lane_e >> 4, r16 = lane_e & 15;
            const int row0 = seq * L;
            const int ncol = pn * 384 + wave * 48 + 4 * q;            // + 16 c
            float* xo = reinterpret_cast<float*>(p.out);
            auto row_ok = [&](int rb) { return rb * 16 + r16 < L; };      // rows beyond the sequence belong to the next tile
            auto row_of = [&](int rb) {                                   // (clamped: loads of rows that are not stored stay in bounds)
                const int r = rb * 16 + r16;
                return row0 + (r < L ? r : L - 1);
            };
            // rolling window of residual rows: four row blocks in flight (the fragment registers of the K loop are dead)
            constexpr int WIN = 4;
            f32x4 xw[IN_HL ? 1 : WIN][IN_HL ? 1 : NCB];
            u32x2 hw[IN_HL ? WIN : 1][IN_HL ? NCB : 1], lw[IN_HL ? WIN : 1][IN_HL ? NCB : 1];
            auto win_load = [&](int rb) {
                const size_t m = (size_t)row_of(rb);
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    if constexpr (IN_HL) {
                        hw[rb % WIN][c] = *reinterpret_cast<const u32x2*>(p.out2 + m * p.ld2 + ncol + 16 * c);
                        lw[rb % WIN][c] = *reinterpret_cast<const u32x2*>(p.lo + m * p.N + ncol + 16 * c);
                    } else {
                        xw[rb % WIN][c] = *reinterpret_cast<const f32x4*>(xo + m * p.ldc + ncol + 16 * c);
                    }
                }
            };
#pragma unroll
            for (int rb = 0; rb < WIN; ++rb) win_load(rb);
            // the next item's first K-tile (older than the window's loads) has landed once at most those loads are outstanding
            wait_vm<WIN * NCB * (IN_HL ? 2 : 1)>();
            f32x4 bv[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) bv[c] = *reinterpret_cast<const f32x4*>(smem + GS_BIAS + (wave * 48 + 16 * c + 4 * q) * 4);
            const int sg = pn * 8 + wave;                             // column group of this wave (48 columns)
            u32x2 h_prev[NCB], l_prev[OUT_HL ? NCB : 1];              // even row block of a pair, waiting for its partner
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int m = row_of(rb);
                const bool ok = row_ok(rb);
                const float mu_r = *reinterpret_cast<const float*>(smem + GS_MU + (rb * 16 + r16) * 4);
                float muc_r = 0.f;
                if constexpr (IN_HL) muc_r = *reinterpret_cast<const float*>(smem + GS_MUC + (rb * 16 + r16) * 4);
                f32x4 v[NCB];
                float sum = 0.f;
                u32x2 h_cur[NCB], l_cur[OUT_HL ? NCB : 1];
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    f32x4 xin;
                    if constexpr (IN_HL) {
                        const half4 hh = __builtin_bit_cast(half4, hw[rb % WIN][c]), ll = __builtin_bit_cast(half4, lw[rb % WIN][c]);
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) xin[e4] = (muc_r + (float)hh[e4]) + (float)ll[e4];
                    } else {
                        xin = xw[rb % WIN][c];
                    }
                    v[c] = xin + (acc[rb][c] + bv[c]);
                    sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
                    if constexpr (!OUT_HL) {
                        if (ok) *reinterpret_cast<f32x4*>(xo + (size_t)m * p.ldc + ncol + 16 * c) = v[c];
                    }
                    half4 h4, l4;
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float d = v[c][e4] - mu_r;
                        const half_t hh = (half_t)d;
                        h4[e4] = hh;
                        if constexpr (OUT_HL) l4[e4] = (half_t)(d - (float)hh);
                    }
                    h_cur[c] = __builtin_bit_cast(u32x2, h4);
                    if constexpr (OUT_HL) l_cur[c] = __builtin_bit_cast(u32x2, l4);
                }
                if (rb + WIN < RB) win_load(rb + WIN);
                // statistics of the row over this wave's 48 columns: (sum, sum of squared deviations from the group mean)
                sum = sum_rows(sum);
                const float gm = sum * (1.0f / 48.0f);
                float m2 = 0.f;
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        const float d = v[c][e4] - gm;
                        m2 = fmaf(d, d, m2);
                    }
                m2 = sum_rows(m2);
                if (q == 0 && ok) *reinterpret_cast<f32x2*>(p.stats + ((size_t)m * p.stats_ld + sg) * 2) = f32x2{sum, m2};
                // fp16 copy (and lo): row blocks rb - 1 (even) and rb (odd) are paired through v_permlane16_swap - even 16-lane
                // groups end up with 8 consecutive columns of the even block's row, odd groups with 8 of the odd block's row
                if (rb & 1) {
                    const int rbs = (q & 1) ? rb : rb - 1;
                    const int r = rbs * 16 + r16;
                    const size_t ms = (size_t)row0 + (r < L ? r : L - 1);
                    const bool oks = r < L;
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        const int nb = pn * 384 + wave * 48 + 16 * c + 4 * (q & ~1);
                        {
                            const auto s0 = __builtin_amdgcn_permlane16_swap(h_prev[c][0], h_cur[c][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(h_prev[c][1], h_cur[c][1], false, false);
                            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                            if (oks) *reinterpret_cast<u32x4*>(p.out2 + ms * p.ld2 + nb) = o;
                        }
                        if constexpr (OUT_HL) {
                            const auto s0 = __builtin_amdgcn_permlane16_swap(l_prev[c][0], l_cur[c][0], false, false);
                            const auto s1 = __builtin_amdgcn_permlane16_swap(l_prev[c][1], l_cur[c][1], false, false);
                            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                            if (oks) *reinterpret_cast<u32x4*>(p.lo + ms * p.N + nb) = o;
                        }
                    }
                } else if (rb == RB - 1) {      // the last row block has no partner: 8-byte stores
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        if (ok) *reinterpret_cast<u32x2*>(p.out2 + (size_t)m * p.ld2 + ncol + 16 * c) = h_cur[c];
                        if constexpr (OUT_HL) {
                            if (ok) *reinterpret_cast<u32x2*>(p.lo + (size_t)m * p.N + ncol + 16 * c) = l_cur[c];
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        h_prev[c] = h_cur[c];
                        if constexpr (OUT_HL) l_prev[c] = l_cur[c];
                    }
                }
            }
        }
        // the tables and stages 1 / 2 are free for the next item once every wave is here; its first K-tile has landed (above)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        barrier_raw();
        if (!has_next) break;
        e = e_n;
        seq = seq_n;
        pn = pn_n;
    }
#endif
}

// ---- weight packing (load time): W [N, K] fp16 -> Wp[N / 384][K / 32][wave][c][lane][8] (hg_seq_dev.h)
__global__ __launch_bounds__(256) void pack_seq_kernel(const half_t* __restrict__ W, half_t* __restrict__ Wp, int N, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece: (pn, k32, wave, c, lane)
    const int K32 = K / 32;
    const size_t total = (size_t)(N / 384) * K32 * 8 * SQ_NCB * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    size_t f = i >> 6;
    const int c = (int)(f % SQ_NCB); f /= SQ_NCB;
    const int wave = (int)(f & 7); f >>= 3;
    const int k32 = (int)(f % K32);
    const int pn = (int)(f / K32);
    const int n = pn * 384 + wave * 48 + c * 16 + (lane & 15);
    *reinterpret_cast<half8*>(Wp + i * 8) = *reinterpret_cast<const half8*>(W + (size_t)n * K + 32 * k32 + 8 * (lane >> 4));
}
hipError_t launch_pack_seq(const half_t* W, half_t* Wp, int N, int K, hipStream_t s) {
    if (N % 384 || K % 32 || !W || !Wp) return hipErrorInvalidValue;
    const size_t total = (size_t)(N / 384) * (K / 32) * 8 * SQ_NCB * 64;
    hipLaunchKernelGGL(pack_seq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, Wp, N, K);
    return hipGetLastError();
}

bool gemm_seq_ok(const GemmArgs& a, int n_seq, int L) {
    if (n_seq < 1 || L <= 192 || L > SQ_RB * 16 || a.M != n_seq * L) return false;
    if (a.N % 384 || a.N > 8 * 384 || a.K % 192 || a.K < 384) return false;      // stage of K-tile kt = kt % 3; >= 6 K-tiles
    if (a.lda < a.K || (a.lda & 7) || a.ldc < a.N || (a.ldc & 3)) return false;
    if (!a.out2 || !a.stats || !a.mu || a.stats_ld != 8 * (a.N / 384)) return false;
    if (a.hl < 0 || a.hl > 3 || (a.hl && (!a.lo || ((a.hl == 2 || a.hl == 3) && !a.muc)))) return false;
    const int ld2 = a.ld2 ? a.ld2 : a.ldc;
    if (ld2 < a.N || (ld2 & 7)) return false;
    const size_t Mp = (size_t)((a.M + 255) / 256) * 256;
    if (Mp * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;
    return true;
}

template <int HL>
static hipError_t launch_seq_t(const GemmArgs& a, const half_t* wp, int n_seq, int L, hipStream_t s) {
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_seq_kernel<HL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int n_items = n_seq * (a.N / 384);
    int grid = n_cu_d[dev_i] & ~7;                 // XCD-wise dealing wants a multiple of 8
    if (grid < 8) grid = n_cu_d[dev_i];
    if (n_items < grid) grid = n_items;            // (not a multiple of 8: plain dealing)
    const unsigned a_bytes = (unsigned)((size_t)((a.M + 255) / 256) * 256 * a.lda * 2);
#ifdef HG_EXPERIMENTS
    static const int mode = []() { const char* e = getenv("HG_GS_MODE"); return e ? atoi(e) : 0; }();
#else
    constexpr int mode = 0;
#endif
    hipLaunchKernelGGL((gemm_seq_kernel<HL>), dim3(grid), dim3(512), GS_LDS, s, a, wp, n_seq, L, a_bytes, mode);
    return hipGetLastError();
}

hipError_t launch_gemm_seq(const GemmArgs& a_in, const half_t* wp, int n_seq, int L, hipStream_t s) {
    GemmArgs a = a_in;
    if (!a.ld2) a.ld2 = a.ldc;
    if (!wp || !gemm_seq_ok(a, n_seq, L)) return hipErrorInvalidValue;
    switch (a.hl) {
        case 1: return launch_seq_t<1>(a, wp, n_seq, L, s);
        case 2: return launch_seq_t<2>(a, wp, n_seq, L, s);
        case 3: return launch_seq_t<3>(a, wp, n_seq, L, s);
        default: return launch_seq_t<0>(a, wp, n_seq, L, s);
    }
}

}  // namespace hg
