// LayerNorm-folded GEMMs with fp16 output on sequence tiles (gfx950): the MLP's c_fc + QuickGELU of a vision-tower block
// (clipnet/model.py:162-164,185-188), EPI_LN_BIAS_QGELU_F16 / EPI_LN_BIAS_F16 of hg_gemm_ring.hip, on the K loop of hg_seq_dev.h
// (the loop of the fused in_proj + attention kernel, hg_qkv_attn.hip):
//
//   out[m][n] = f( rstd[m] * (x16[m][:] W'[n][:]^T - (mean[m] - c[m]) * cs[n]) + b'[n] ),   f = QuickGELU or identity
//
// A work item is (sequence, 384-column panel): a 208 x 384 tile whose rows are ONE sequence.  N = 3072 gives 8 panels, 2 048 items
// at batch 256 = exactly eight per CU (the 256 x 256 ring needs 9.23 rounds).  This is the regime the loop is good in: the
// activations (the 77 MB centred fp16 copy of the stream, just written) and a 590 KB weight panel are hot in L2, K = 768 - the
// in_proj runs it at 1.58 PFLOP/s in the loop.  (The residual GEMMs do not pay on it: tools/experiments/round4/
// README_gemm_seq_rln.md.)  XCD x owns the sequences [x * spx, (x + 1) * spx); its list is panel-group major (groups of gsz
// panels), sequence next, panel fastest: the panels of a group run side by side on a sequence's rows.
// The epilogue uses the expressions of hg_gemm_ring.hip's f16 epilogue (bit-identical results: tests/test_gpu_gemm.py); rows
// leave as 16-byte pieces, two row blocks paired through v_permlane16_swap.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_seq_dev.h"

namespace hg {

namespace {
constexpr int GS_BIAS = SQ_END;                    // bias' of the panel [384]
constexpr int GS_CS = GS_BIAS + 384 * 4;           // column sums of the folded weight [384]
constexpr int GS_MR = GS_CS + 384 * 4;             // (mean - c, rstd) of the sequence's rows [208][2]
constexpr int GS_LDS = GS_MR + SQ_RB * 16 * 8;
static_assert(GS_LDS <= 160 * 1024, "LDS budget");
}  // namespace

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_seq_kernel(const GemmArgs p, const half_t* __restrict__ wp, const int n_seq, const int L,
                                                          const unsigned a_bytes, const int gsz, const int stagger, const int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    // timing-experiment switches (HG_GS_MODE bits: 2 no MFMA in the K loop, 4 no epilogue, 8 no operand DMA; wrong results) exist only
    // in a -DHG_EXPERIMENTS build
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int RB = SQ_RB, NCB = SQ_NCB;
    constexpr bool GELU = EPI == EPI_LN_BIAS_QGELU_F16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K >> 6;
    const int PN = p.N / 384;

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
    // De-phased epilogues: every item takes the same time and every CU owns the same number, so all CUs would store their 160 KB
    // tile at the same moment - HBM idles during the K loops and saturates during the epilogues, and vmcnt (in issue order) makes
    // the next item's first operand wait stand behind the whole burst.  The workgroups of consecutive sequences start `stagger >> 4`
    // cycles apart in `stagger & 15` phases (workgroups on the panels of one sequence stay in step: they share its rows in L2).
    if ((stagger & 15) > 1) {
        const int ph = (idx / gsz) % (stagger & 15);
        const long long d = (long long)ph * (stagger >> 4);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)(__builtin_amdgcn_s_memtime() - t0) < d) __builtin_amdgcn_s_sleep(32);
    }
    auto decode = [&](int e, int& seq, int& pn) {
        const int per = ns * gsz;
        const int grp = e / per, rem = e - grp * per;
        const int s = rem / gsz;
        seq = s0 + s;
        pn = grp * gsz + (rem - s * gsz);
    };

#define SQ_A_PTR p.A
#define SQ_A_BYTES a_bytes
#define SQ_LDA p.lda
#define SQ_W_PTR wp
#define SQ_W_BYTES (unsigned)((size_t)p.N * p.K * 2)
#include "hg_seq_kloop.inc"

    // bias' and cs of the panel (2 x 1.5 KiB: waves 0-3), (mean - c, rstd) of the sequence's rows (208 x 8 B: waves 4, 5)
    auto issue_extras = [&](int seq, int pn) {
        if (wave < 4) {
            const float* src = (wave & 2) ? p.cs : p.bias;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(p.N * 4), 0x00020000);
            char* dst = smem + ((wave & 2) ? GS_CS : GS_BIAS) + (wave & 1) * 1024;
            if (!(wave & 1) || lane < 32)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (HG_LDS void*)dst, 16, lane * 16, pn * 384 * 4 + (wave & 1) * 1024, 0, 0);
        } else if (wave < 6) {
            const __amdgpu_buffer_rsrc_t rsM =
                __builtin_amdgcn_make_buffer_rsrc((void*)p.mr, 0, (unsigned)((size_t)n_seq * L * 8), 0x00020000);
            if (wave == 4 || lane < (RB * 16 * 8 - 1024) / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + GS_MR + (wave - 4) * 1024), 16, lane * 16,
                                                         seq * L * 8 + (wave - 4) * 1024, 0, 0);
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
        // the next item's first K-tile has landed before this item's stores enter vmcnt
        wait_vm<0>();

        if (xmode & 4) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int c = 0; c < NCB; ++c) asm volatile("" ::"v"(acc[rb][c]));
        } else {
            // ---------------- epilogue (opaque lane id: its lane constants must not be hoisted above the K loop, where every register is taken)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int q = lane_e >> 4, r16 = lane_e & 15;
            const int row0 = seq * L;
            half_t* outp = reinterpret_cast<half_t*>(p.out);
            f32x4 gk = {0.f, 0.f, 0.f, 0.f};
            if constexpr (GELU) gk = quick_gelu_consts();
            (void)gk;
            f32x4 bv[NCB], cv[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int col = wave * 48 + 16 * c + 4 * q;
                bv[c] = *reinterpret_cast<const f32x4*>(smem + GS_BIAS + col * 4);
                cv[c] = *reinterpret_cast<const f32x4*>(smem + GS_CS + col * 4);
            }
            auto cvt2 = [](float a, float b) {      // RNE, one v_cvt_pk_f16_f32
                const half2v h = __builtin_convertvector(f32x2{a, b}, half2v);
                return __builtin_bit_cast(unsigned, h);
            };
            unsigned d_prev[NCB][2];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const f32x2 mr = *reinterpret_cast<const f32x2*>(smem + GS_MR + (rb * 16 + r16) * 8);
                unsigned d_cur[NCB][2];
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    f32x4 v = (acc[rb][c] - cv[c] * mr[0]) * mr[1] + bv[c];      // rstd * (acc - mean * cs) + bias'
                    if constexpr (GELU) v = quick_gelu4(v, gk);
                    d_cur[c][0] = cvt2(v[0], v[1]);
                    d_cur[c][1] = cvt2(v[2], v[3]);
                }
                // row blocks rb - 1 (even) and rb (odd) are paired through v_permlane16_swap: even 16-lane groups end up with 8
                // consecutive columns of the even block's row, odd groups with 8 of the odd block's row - 16-byte stores
                if (rb & 1) {
                    const int r = ((q & 1) ? rb : rb - 1) * 16 + r16;
                    half_t* rowp = outp + (size_t)(row0 + r) * p.ldc + pn * 384 + wave * 48 + 4 * (q & ~1);
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        const auto s0 = __builtin_amdgcn_permlane16_swap(d_prev[c][0], d_cur[c][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(d_prev[c][1], d_cur[c][1], false, false);
                        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
                        if (r < L) *reinterpret_cast<u32x4*>(rowp + 16 * c) = o;
                    }
                } else if (rb == RB - 1) {      // the last row block has no partner: 8-byte stores
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const int r = rb * 16 + r16;
                    half_t* rowp = outp + (size_t)(row0 + r) * p.ldc + pn * 384 + wave * 48 + 4 * q;
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        if (r < L) *reinterpret_cast<u32x2*>(rowp + 16 * c) = u32x2{d_cur[c][0], d_cur[c][1]};
                } else {
#pragma unroll
                    for (int c = 0; c < NCB; ++c) { d_prev[c][0] = d_cur[c][0]; d_prev[c][1] = d_cur[c][1]; }
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

bool gemm_seq_ok(int epi, const GemmArgs& a, int n_seq, int L) {
    if (epi != EPI_LN_BIAS_QGELU_F16 && epi != EPI_LN_BIAS_F16) return false;
    if (n_seq < 1 || L <= 192 || L > SQ_RB * 16 || a.M != n_seq * L) return false;
    if (a.N % 384 || a.K % 192 || a.K < 384) return false;      // stage of K-tile kt = kt % 3; >= 6 K-tiles
    if (a.lda < a.K || (a.lda & 7) || a.ldc < a.N || (a.ldc & 7)) return false;
    if (!a.cs || !a.mr || !a.bias || !a.out || !a.A) return false;
    const size_t Mp = (size_t)((a.M + 255) / 256) * 256;
    if (Mp * a.lda * 2 >= (1ull << 31) || (size_t)a.N * a.K * 2 >= (1ull << 31)) return false;
    return true;
}

template <int EPI>
static hipError_t launch_seq_t(const GemmArgs& a, const half_t* wp, int n_seq, int L, int gsz, int stagger, hipStream_t s) {
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_seq_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int PN = a.N / 384;
    if (gsz <= 0 || PN % gsz) gsz = PN;
    const int n_items = n_seq * PN;
    int grid = n_cu_d[dev_i] & ~7;                 // XCD-wise dealing wants a multiple of 8
    if (grid < 8) grid = n_cu_d[dev_i];
    if (n_items < grid) grid = n_items;            // (not a multiple of 8: plain dealing)
    const unsigned a_bytes = (unsigned)((size_t)((a.M + 255) / 256) * 256 * a.lda * 2);
#ifdef HG_EXPERIMENTS
    static const int mode = []() { const char* e = getenv("HG_GS_MODE"); return e ? atoi(e) : 0; }();
#else
    constexpr int mode = 0;
#endif
    hipLaunchKernelGGL((gemm_seq_kernel<EPI>), dim3(grid), dim3(512), GS_LDS, s, a, wp, n_seq, L, a_bytes, gsz, stagger, mode);
    return hipGetLastError();
}

hipError_t launch_gemm_seq(int epi, const GemmArgs& a, const half_t* wp, int n_seq, int L, int gsz, int stagger, hipStream_t s) {
    if (!wp || !gemm_seq_ok(epi, a, n_seq, L)) return hipErrorInvalidValue;
    return epi == EPI_LN_BIAS_QGELU_F16 ? launch_seq_t<EPI_LN_BIAS_QGELU_F16>(a, wp, n_seq, L, gsz, stagger, s)
                                        : launch_seq_t<EPI_LN_BIAS_F16>(a, wp, n_seq, L, gsz, stagger, s);
}

}  // namespace hg
