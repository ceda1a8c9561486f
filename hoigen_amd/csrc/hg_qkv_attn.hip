// Fused in_proj (QKV projection, LayerNorm folded in) + scaled-dot-product attention for gfx950: q, k and v never leave
// the chip.  Replaces the pair  gemm_ring<EPI_LN_BIAS_F16> -> attention_kernel  of a vision-tower block
// (clipnet/model.py:171,181-183: nn.MultiheadAttention's in_proj + SDPA) when a sequence fits one
// row tile (192 < L <= 208: the 197 tokens of ViT-B/16).  Per layer at batch 256 the separate kernels write 232 MB of qkv and
// read it back; here the only HBM traffic is the operand stream (77 MB of activations, the weight from L2) and the
// 77 MB attention output.
//
// Work item = (sequence, head PAIR): a 208 x 384 output tile [q_a | k_a | v_a | q_b | k_b | v_b] (13 row blocks x 24 column
// blocks of 16), K = D.  1 536 items at batch 256 = six per CU, dealt XCD-wise so that the head pairs of a sequence run side by
// side on one XCD (its activation panel is fetched once and shared through that XCD's L2).
//
// GEMM phase (the row-owner structure of tools/experiments/round4/hg_gemm_rows.hip with the rows of ONE sequence):
//   8 waves, all along N: wave w owns column blocks 3w .. 3w+2 of every row: 13 x 3 accumulator blocks = 156 VGPRs;
//   waves 0-3 hold head a, waves 4-7 head b.
//   A (centred fp16 copy of the stream): shared ring of three K-tile stages (208 rows x 128 B, XOR-swizzled 16-byte chunks as
//     in hg_gemm_ring.hip), buffer_load ... lds, three K-tiles ahead; every wave reads every row.
//   W: packed once at load time into MFMA-fragment order (pack_qkv_kernel), streamed from L2 into WAVE-PRIVATE rings of one
//     K-tile (2 x 3 fragments of 1 KiB, read with ds_read_b128 at lane * 16: no swizzle, no sharing, no barrier); a slot is
//     refilled with the next K-tile's fragment as soon as its 13 MFMAs are issued.
//   One s_barrier per K-tile (five fragment reads before its end, so that no wave drains at the boundary), counted vmcnt
//   throughout; K-tiles are instantiated by position (first / middle / last) because the VMEM sequence differs there (below).  Bytes through the CU's load path per K-tile: 26 KiB of A +
//   48 KiB of W for 2 x 13 x 24 MFMAs = 7.4 KB per MFLOP (the 256 x 256 ring: 7.6, the 128 x 256 ring2: 11.4).
// Epilogue: rstd * (acc - (mean - c) * cs) + b' as in the EPI_LN_BIAS_F16 epilogue of hg_gemm_ring.hip (same expression,
//   same rounding to fp16), written to LDS as Q, K, V rows of 128 B in the layout attention_kernel stages them in.
// Attention phase: the tile functions of hg_attn.hip (hg_attn_dev.h) on those rows - head a first (waves 4-7 keep head b as
//   packed fp16 in 78 registers meanwhile), then head b; one wave per 32-query tile, the wave's output tile leaves through its
//   own (dead) Q rows as whole 128-byte lines.  Bit-identical to the separate kernels (tests/test_gpu_attention.py).
//
// LDS (162 432 B): W rings 48 KiB | A stage 0 26 KiB | 80 KiB that hold A stages 1 and 2 during the K loop and Q, K, V of one
// head during the attention phases | bias' and column sums of the pair (3 KiB) | (mean - c, rstd) of the 208 rows.  Stage 0
// and the W rings are NOT touched by the attention phases: the first K-tile of the next item is fetched under the last
// K-tiles of this one and waits there; its stages 1 and 2 are issued when the attention phases have released the region.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_attn_dev.h"
#ifdef HG_STAMPS      // s_memtime brackets around the K loop's waits (hg_seq_kloop_run.inc): totals in tks[], reported as stamps 12-14
#define SQ_STAMP_B() do { tk_b = __builtin_amdgcn_s_memtime(); } while (0)
#define SQ_STAMP_E(k) do { tks[k] += __builtin_amdgcn_s_memtime() - tk_b; } while (0)
#endif
#include "hg_seq_dev.h"

namespace hg {

namespace {
constexpr int QA_RB = SQ_RB;                       // 16-row blocks of a sequence tile
constexpr int QA_NCB = SQ_NCB;                     // 16-column blocks per wave
constexpr int QA_ASTG = SQ_ASTG;                   // one A stage = one Q / K / V matrix: 208 rows x 128 B
[[maybe_unused]] constexpr int QA_ATT = SQ_S12;                     // stages 1 and 2 of the K loop = the attention operands
constexpr int QA_ATT_BYTES = SQ_S12_BYTES;         // 3 x QA_ASTG + slack (tile 6 of Q / K reads 16 rows into the next matrix)
constexpr int QA_BCS = SQ_END;                     // bias'[384] | cs[384] in tile column order
constexpr int QA_MR = QA_BCS + 2 * 384 * 4;
constexpr int QA_LDS = QA_MR + QA_RB * 16 * 8;
static_assert(QA_LDS <= 160 * 1024, "LDS budget");
static_assert(3 * QA_ASTG + 16 * 128 <= QA_ATT_BYTES, "attention operands");
}  // namespace

__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(const QkvAttnArgs p, const int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    // timing-experiment switches (HG_QA_MODE bits: 1 no attention phases, 2 no MFMA in the K loop, 4 no epilogue at all,
    // 8 no operand DMA; wrong results) exist only in a -DHG_EXPERIMENTS build
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    constexpr int RB = QA_RB, NCB = QA_NCB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.D >> 6;                       // K-tiles per item (a multiple of 3: stage of K-tile kt = kt % 3)
    const int HP = p.heads >> 1;

    // ---- this workgroup's items.  XCD x (= blockIdx % 8 under round-robin placement; speed only) owns the sequences
    // [x * spx, (x + 1) * spx); its list is head-pair-group major (groups of gsz pairs), sequence next, pair fastest, and its
    // workgroups walk it cpx items at a time: with gsz = HP the pairs of a sequence run side by side.
    const int G = gridDim.x, bid = blockIdx.x;
    const bool xcd_ok = (G & 7) == 0;
    const int cpx = xcd_ok ? (G >> 3) : G;
    const int idx = xcd_ok ? (bid >> 3) : bid;
    const int spx = xcd_ok ? ((p.n_seq + 7) >> 3) : p.n_seq;
    const int s0 = xcd_ok ? (bid & 7) * spx : 0;
    int ns = p.n_seq - s0;
    ns = ns < 0 ? 0 : (ns > spx ? spx : ns);
    const int nx = ns * HP;
    if (idx >= nx) return;
    const int gsz = p.gsz;
    auto decode = [&](int e, int& seq, int& hp) {
        const int per = ns * gsz;
        const int grp = e / per, rem = e - grp * per;
        const int s = rem / gsz;
        seq = s0 + s;
        hp = grp * gsz + (rem - s * gsz);
    };

#define SQ_A_PTR p.x16
#define SQ_A_BYTES p.a_bytes
#define SQ_LDA p.lda
#define SQ_W_PTR p.wp
#define SQ_W_BYTES (unsigned)((size_t)3 * p.D * p.D * 2)
#include "hg_seq_kloop.inc"
    // bias' | cs of the pair (3 KiB: waves 0-2) and (mean - c, rstd) of the sequence's rows (208 x 8 B: waves 3 and 4)
    auto issue_extras = [&](int seq, int hp) {
        // (descriptors built here, once per item: they would otherwise sit in 8 SGPRs through the K loop)
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bcs, 0, (unsigned)(HP * 768 * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsM =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.mr, 0, (unsigned)((size_t)p.n_seq * p.L * 8), 0x00020000);
        if (wave < 3)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (HG_LDS void*)(smem + QA_BCS + wave * 1024), 16, lane * 16,
                                                     hp * 768 * 4 + wave * 1024, 0, 0);
        else if (wave == 3)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + QA_MR), 16, lane * 16, seq * p.L * 8, 0, 0);
        else if (wave == 4) {
            if (lane < (RB * 16 * 8 - 1024) / 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (HG_LDS void*)(smem + QA_MR + 1024), 16, lane * 16,
                                                         seq * p.L * 8 + 1024, 0, 0);
        }
    };

    // ---- prologue: K-tile 0 of the first item
    int e = idx, seq, hp;
    decode(e, seq, hp);
    seq_prologue(seq * p.L, hp);

#ifdef HG_STAMPS
    unsigned long long tks[3] = {0, 0, 0}, tk_b = 0;
    unsigned long long tst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = __builtin_amdgcn_s_memtime(), t_all0 = t_prev;
#define QA_ST(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tst[k] += t_ - t_prev; t_prev = t_; } while (0)
#else
#define QA_ST(k) do {} while (0)
#endif
    for (;;) {
        const int e_n = e + cpx;
        const bool has_next = e_n < nx;
        int seq_n = seq, hp_n = hp;          // no next item: the run-ahead loads fetch this item's first K-tile again (never read)
        if (has_next) decode(e_n, seq_n, hp_n);

        // the attention phases of the previous item have released the 80 KiB: the epilogue's tables, then the K loop (stages 1, 2)
        issue_extras(seq, hp);
        f32x4 acc[RB][NCB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int c = 0; c < NCB; ++c) acc[rb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const int sq_row0 = seq * p.L, sq_pn = hp, sq_row0_n = seq_n * p.L, sq_pn_n = hp_n;
#include "hg_seq_kloop_run.inc"
        }
        QA_ST(0);      // K loop
        wait_vm<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        barrier_raw();
        QA_ST(1);      // drain + barrier

        if (xmode & 4) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int c = 0; c < NCB; ++c) asm volatile("" ::"v"(acc[rb][c]));
        } else {
            // ---------------- epilogue: LayerNorm fold, fp16 (the expressions of hg_gemm_ring.hip's EPI_LN_BIAS_F16 epilogue)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            // (an opaque copy of the lane id: the lane constants of these phases are recomputed per item instead of being hoisted
            // above the K loop, where every register is taken)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int q = lane_e >> 4, r16 = lane_e & 15;
            unsigned held[RB][NCB][2];
            {
                f32x4 bv[NCB], cv[NCB];
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const int col = (wave * NCB + c) * 16 + 4 * q;
                    bv[c] = *reinterpret_cast<const f32x4*>(smem + QA_BCS + col * 4);
                    cv[c] = *reinterpret_cast<const f32x4*>(smem + QA_BCS + 384 * 4 + col * 4);
                }
                auto cvt2 = [](float a, float b) {      // RNE, one v_cvt_pk_f16_f32
                    const half2v h = __builtin_convertvector(f32x2{a, b}, half2v);
                    return __builtin_bit_cast(unsigned, h);
                };
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const f32x2 mr = *reinterpret_cast<const f32x2*>(smem + QA_MR + (rb * 16 + r16) * 8);
#pragma unroll
                    for (int c = 0; c < NCB; ++c) {
                        const f32x4 v = (acc[rb][c] - cv[c] * mr[0]) * mr[1] + bv[c];      // rstd * (acc - mean * cs) + bias'
                        held[rb][c][0] = cvt2(v[0], v[1]);
                        held[rb][c][1] = cvt2(v[2], v[3]);
                    }
                }
            }
            // this wave's 3 column blocks -> rows of Q, K or V: block l12 = 3 (wave % 4) + c of the head = matrix l12 / 4,
            // columns 16 (l12 % 4) + 4 q ..; 16-byte chunks XOR-swizzled by row as attention_kernel's DMA leaves them
            // (both swizzles have period 16 in the row: one address per column block, the 13 row blocks are immediate offsets)
            auto write_head = [&]() {
                static_assert(QA_RB * 16 * ROWB < 65536, "ds_write offset field");
                const int swk = swz_k(r16), swv = swz_v(r16);
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const int l12 = (wave & 3) * NCB + c;
                    const int mtx = l12 >> 2, sub = l12 & 3;
                    const int chunk = 2 * sub + (q >> 1);
                    char* dst = smem + QA_ATT + mtx * QA_ASTG + (q & 1) * 8 + r16 * ROWB + ((chunk ^ (mtx == 2 ? swv : swk)) << 4);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) {
                        u32x2 hv = u32x2{held[rb][c][0], held[rb][c][1]};
                        // V rows >= L are the NEXT sequence's first rows (or workspace padding): their keys are masked (P = 0), but
                        // 0 * Inf / 0 * NaN in P V would poison every query of THIS sequence - they are stored as zeros
                        // (attention_kernel clamps its pad rows to row L - 1 instead; with finite data both give the same bits)
                        if (rb == RB - 1 && mtx == 2 && (RB - 1) * 16 + r16 >= p.L) hv = u32x2{0u, 0u};
                        *reinterpret_cast<u32x2*>(dst + rb * 16 * ROWB) = hv;
                    }
                }
            };
            // One head: wave w < 7 runs query tile w over the 7 key tiles, then stores its 32 x 64 tile.  The arithmetic per key
            // tile is attention_kernel's (hg_attn_dev.h: tile_scores / tile_softmax_pv, same operations in the same order: the two
            // kernels are bit-identical), but here at most two waves share a SIMD and nothing else hides a wave's dependent chain
            // K read -> S^T MFMAs -> max -> exp -> P V, so the seven tiles are unrolled and software-pipelined by hand: the
            // S^T MFMAs of tile kt+1 and the K fragments of tile kt+2 are issued before the softmax of tile kt, the V fragments
            // of tile kt before its softmax arithmetic (sequence length is fixed here: 7 tiles, the last one masked and half empty).
            auto attend = [&](const int head) {
                if ((xmode & 1) || wave >= 7) return;
                const char* Qs = smem + QA_ATT;
                const char* Ks = Qs + QA_ASTG;
                const char* Vs = Ks + QA_ASTG;
                const int L = p.L;
                constexpr int NKT = 7;
                int lane = lane_e;
                asm volatile("" : "+v"(lane));
                const int qt = wave, qcol = lane & 31, hh = lane >> 5;
                const int qq = qt * 32 + qcol;
                int k_off[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) k_off[ks] = qcol * ROWB + (((2 * ks + hh) ^ swz_k(qcol)) << 4);
                half8 qf[4];       // rows beyond the tile alias K rows (finite, never stored)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const half8*>(Qs + qt * TILEB + k_off[ks]);
                const int gi = lane >> 4, l16 = lane & 15;
                const int vq = l16 >> 2, vp = l16 & 3;
                unsigned v_addr[2];      // LDS byte addresses of this lane's transposing reads in key tile 0
                {
                    const int key0 = 4 * (gi >> 1) + vq;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int chunk = dt * 4 + (gi & 1) * 2 + (vp >> 1);
                        v_addr[dt] = (unsigned)(size_t)(HG_LDS const char*)(Vs + key0 * ROWB + ((chunk ^ swz_v(key0)) << 4) + (vp & 1) * 8);
                    }
                }
                float cexp = 0.125f * 1.4426950408889634f;         // head_dim^-0.5 * log2(e)
                asm volatile("" : "+v"(cexp));                     // (made here, per call: not a value to keep across the K loop)
                float m = -1.0e30f, lsum = 0.f;
                f32x16 o[2];
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
                half8 kf[2][4];
                f32x16 sc[2];
                auto load_k = [&](int kt, int b) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) kf[b][ks] = *reinterpret_cast<const half8*>(Ks + kt * TILEB + k_off[ks]);
                };
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                auto scores = [&](int b) {      // = tile_scores() (the first MFMA takes the zero accumulator as an inline constant)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        sc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[b][ks], qf[ks], ks == 0 ? zero16 : sc[b], 0, 0, 0);
                };
                auto max3 = [](float a, float b, float c) {      // (fmaxf would canonicalise every input: twice the instructions)
                    float d;
                    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
                    return d;
                };
                // An in-order wave issues nothing behind an MFMA that waits for the matrix pipe, so the four S^T MFMAs of the NEXT
                // tile are spread over this tile's exponentials (four elements between two MFMAs): MFMA and VALU of ONE wave overlap
                // (sched_barrier pins the order).
                load_k(0, 0);
                load_k(1, 1);
                scores(0);
                load_k(2, 0);
                if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // the younger wave of a SIMD loses every arbitration otherwise
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const int cur = kt & 1, nxt = cur ^ 1;
                    const bool more = kt + 1 < NKT;
                    // K fragments of tile kt + 2 (the MFMAs that read this buffer were issued a tile ago) and
                    // V fragments of this tile (element j of lane half hh is key 16 s + 8 (j >> 2) + 4 hh + (j & 3)): inline asm, the
                    // compiler's waitcnt pass gives the builtin no memory operand; waited for right before the P V MFMAs
                    const bool two_steps = kt * 32 + 16 < RB * 16;      // keys beyond the staged rows (all masked)
                    fp16x4_t vr[2][2][2];
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        if (st == 1 && !two_steps) break;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {      // (tile and step offset as immediates: < 64 KiB)
                            // (the second read, keys + 8: swz_v flips bit 1 of the chunk index there = bit 5 of the byte address)
                            asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%5"
                                         : "=&v"(vr[st][dt][0]), "=&v"(vr[st][dt][1])
                                         : "v"(v_addr[dt]), "v"(v_addr[dt] ^ 32u), "n"(kt * TILEB + st * (16 * ROWB)),
                                           "n"(kt * TILEB + st * (16 * ROWB) + 1024)
                                         : "memory");
                        }
                    }
                    if (kt + 2 < NKT && kt > 0) load_k(kt + 2, cur);
                    // ---- tile_softmax_pv(): mask, running max / rescale, P
                    f32x16& s = sc[cur];
                    // (a tile whose keys + 16 .. + 31 lie beyond the staged rows - the last one - only has its elements r < 8: the others
                    // are masked for every L, their exponentials are +0 and their P columns unused; leaving them out changes no bit)
                    constexpr int NR = 16;
                    const int nr = two_steps ? NR : NR / 2;
                    if (kt * 32 + 32 > RB * 16 - 15) {      // (tiles 0-5 lie inside every L > 192; tile 6 always needs the mask)
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            if (r < nr) {
                                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                                s[r] = key < L ? s[r] : -INFINITY;
                            }
                        }
                    }
                    float mx = max3(max3(s[0], s[1], s[2]), max3(s[3], s[4], s[5]), max3(s[6], s[7], s[7]));
                    if (two_steps) mx = max3(mx, max3(s[8], s[9], s[10]), max3(max3(s[11], s[12], s[13]), s[14], s[15]));
                    {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx),
                                                                         false, false);
                        mx = max3(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]), mx);
                    }
                    if (__any(mx > m)) {                 // some query's running max grew: rescale (wave-uniform branch)
                        const float mn = max3(m, mx, mx);
                        const float alpha = __builtin_amdgcn_exp2f((m - mn) * cexp);
                        m = mn;
                        lsum *= alpha;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                    }
                    const float mc = m * cexp;
                    half8 pf[2];
                    float ps = 0.f;      // (summed in tile_softmax_pv's order: r = 0 .. 15)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                        for (int r = 4 * g4; r < 4 * g4 + 4; ++r) {
                            if (r >= nr) continue;
                            const float ex = __builtin_amdgcn_exp2f(fmaf(s[r], cexp, -mc));
                            ps += ex;
                            pf[r >> 3][r & 7] = (half_t)ex;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (more) sc[nxt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[nxt][g4], qf[g4], g4 == 0 ? zero16 : sc[nxt], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // ---- O^T[d][q] += sum_key V[key][d] P[q][key]
                    if (two_steps)
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0][0]), "+v"(vr[0][0][1]), "+v"(vr[0][1][0]), "+v"(vr[0][1][1]),
                                     "+v"(vr[1][0][0]), "+v"(vr[1][0][1]), "+v"(vr[1][1][0]), "+v"(vr[1][1][1])::"memory");
                    else
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0][0]), "+v"(vr[0][0][1]), "+v"(vr[0][1][0]), "+v"(vr[0][1][1])::"memory");
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            if (st == 0 || two_steps) {
                                half8 vf;
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4) {
                                    vf[e4] = (half_t)vr[st][dt][0][e4];
                                    vf[4 + e4] = (half_t)vr[st][dt][1][e4];
                                }
                                o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[st], o[dt], 0, 0, 0);
                            }
                        }
                    }
                    lsum += ps;
                }
                if (wave >= 4) __builtin_amdgcn_s_setprio(0);
                {      // + the other half's partial sum (lane ^ 32): each lane's pair is (own, other) in one order or the other
                    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lsum), __builtin_bit_cast(unsigned, lsum),
                                                                     false, false);
                    lsum = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
                }
                const float inv = 1.0f / lsum;
                // the wave's tile leaves through its own Q rows (only this wave read them, into qf) as whole 128-byte lines
                char* ot = smem + QA_ATT + wave * 4096;
                if (qq < RB * 16) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            half4 h;
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) h[e4] = (half_t)(o[dt][g * 4 + e4] * inv);
                            *reinterpret_cast<half4*>(ot + qcol * 128 + (((dt * 4 + g) ^ (qcol & 7)) << 4) + hh * 8) = h;
                        }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const int cr = lane >> 3, cc = lane & 7;
                half_t* const obase = p.out + (size_t)seq * L * p.ldo + head * HD;      // (uniform base + 32-bit lane offset)
#pragma unroll
                for (int rb8 = 0; rb8 < 32; rb8 += 8) {
                    const int row = rb8 + cr, qrow = qt * 32 + row;
                    if (qrow < L) {
                        const half8 v = *reinterpret_cast<const half8*>(ot + row * 128 + ((cc ^ (row & 7)) << 4));
                        *reinterpret_cast<half8*>(obase + (unsigned)(qrow * p.ldo + cc * 8)) = v;
                    }
                }
            };
            QA_ST(9);      // (LayerNorm fold arithmetic)
            if (wave < 4) write_head();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            QA_ST(10);     // (head a -> LDS)
            barrier_raw();
            QA_ST(2);      // barrier behind them
            attend(2 * hp);
            QA_ST(3);      // attention a
            __builtin_amdgcn_s_waitcnt(0xC07F);
            barrier_raw();
            QA_ST(4);      // barrier
            if (wave >= 4) write_head();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            barrier_raw();
            QA_ST(5);      // head b -> LDS + barrier
            attend(2 * hp + 1);
            QA_ST(6);      // attention b
            __builtin_amdgcn_s_waitcnt(0xC07F);
            barrier_raw();
            QA_ST(7);      // barrier
        }
        if (!has_next) break;
        e = e_n;
        seq = seq_n;
        hp = hp_n;
    }
#ifdef HG_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 8 + wave) * 16;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = tst[k];
        d[8] = __builtin_amdgcn_s_memtime() - t_all0;
        d[9] = tst[9];
        d[10] = tst[10];
        d[11] = tks[0];
        d[12] = tks[1];
        d[13] = tks[2];
    }
#endif
#endif
}

// ---- weight packing (load time): the LayerNorm-folded in_proj weight [3D, D] into fragment order, bias' and column sums
// into tile column order.  Tile column block lb = 3 wave + c of head pair hp: head 2 hp + lb / 12, matrix (lb % 12) / 4 (q, k, v),
// columns 16 ((lb % 12) % 4) .. + 15 of that head.
__global__ __launch_bounds__(256) void pack_qkv_kernel(const half_t* __restrict__ W, const float* __restrict__ bias,
                                                       const float* __restrict__ cs, half_t* __restrict__ Wp,
                                                       float* __restrict__ bcs, int D, int heads) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece: (hp, k32, wave, c, lane)
    const int HP = heads / 2, K32 = D / 32;
    const size_t total = (size_t)HP * K32 * 8 * QA_NCB * 64;
    auto src_row = [&](int hp, int lb, int j) {
        const int head = 2 * hp + lb / 12, l12 = lb % 12;
        return (l12 >> 2) * D + head * 64 + (l12 & 3) * 16 + j;
    };
    if (i < total) {
        const int lane = (int)(i & 63);
        size_t f = i >> 6;
        const int c = (int)(f % QA_NCB); f /= QA_NCB;
        const int wave = (int)(f & 7); f >>= 3;
        const int k32 = (int)(f % K32);
        const int hp = (int)(f / K32);
        const int n = src_row(hp, wave * QA_NCB + c, lane & 15);
        *reinterpret_cast<half8*>(Wp + i * 8) = *reinterpret_cast<const half8*>(W + (size_t)n * D + 32 * k32 + 8 * (lane >> 4));
    }
    if (i < (size_t)HP * 384) {
        const int hp = (int)(i / 384), j = (int)(i % 384);
        const int n = src_row(hp, j / 16, j % 16);
        bcs[hp * 768 + j] = bias ? bias[n] : 0.f;
        bcs[hp * 768 + 384 + j] = cs[n];
    }
}

hipError_t launch_pack_qkv(const half_t* W, const float* bias, const float* cs, half_t* Wp, float* bcs, int D, int heads,
                           hipStream_t s) {
    if (heads < 2 || (heads & 1) || D != heads * 64 || !W || !cs || !Wp || !bcs) return hipErrorInvalidValue;
    const size_t total = (size_t)(heads / 2) * (D / 32) * 8 * QA_NCB * 64;
    hipLaunchKernelGGL(pack_qkv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, bias, cs, Wp, bcs, D, heads);
    return hipGetLastError();
}

bool qkv_attn_ok(int n_seq, int L, int D, int heads, int lda) {
    if (n_seq < 1 || heads < 2 || (heads & 1) || D != heads * 64) return false;
    if (L <= 192 || L > QA_RB * 16) return false;                 // seven 32-key tiles, thirteen 16-row blocks
    if ((D / 64) % 3 || D / 64 < 6) return false;                 // stage of K-tile kt = kt % 3; four distinct K-tile kinds
    if (lda < D || (lda & 7)) return false;
    const size_t Mp = (size_t)(((size_t)n_seq * L + 255) / 256) * 256;
    if (Mp * lda * 2 >= (1ull << 31) || (size_t)3 * D * D * 2 >= (1ull << 31)) return false;
    return true;
}

// Whether the one kernel beats the two it replaces at this batch: its time is (items / CUs rounded UP) x one item (37 us), so it
// wants its last round well filled - measured (profiles/r04_qkv_attn.txt, sequences: fused / separate us): 32: 43 / 61, 43: 73 / 80,
// 64: 75 / 77, 86: 109 / 105, 128: 122 / 139, 171: 190 / 174, 192: 191 / 199, 256: 222 / 245.  Results are bit-identical either way.
bool qkv_attn_pays(int n_seq, int heads, int n_cu) {
    if (n_cu <= 0) n_cu = 256;
    const long items = (long)n_seq * (heads / 2);
    const long rounds = (items + n_cu - 1) / n_cu;
    return rounds <= 1 || items * 100 >= rounds * n_cu * 88;
}

hipError_t launch_qkv_attn(const QkvAttnArgs& a_in, hipStream_t s) {
    QkvAttnArgs a = a_in;
    if (!qkv_attn_ok(a.n_seq, a.L, a.D, a.heads, a.lda) || !a.x16 || !a.wp || !a.bcs || !a.mr || !a.out) return hipErrorInvalidValue;
    if (a.ldo <= 0) a.ldo = a.D;
    if (a.ldo < a.D || (a.ldo & 7)) return hipErrorInvalidValue;
    const int HP = a.heads / 2;
    if (a.gsz <= 0 || HP % a.gsz) a.gsz = HP;
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int n_items = a.n_seq * HP;
    int grid = n_cu_d[dev_i] & ~7;                 // XCD-wise dealing wants a multiple of 8
    if (grid < 8) grid = n_cu_d[dev_i];
    if (n_items < grid) grid = n_items;            // (not a multiple of 8: plain dealing)
    if (!a.a_bytes) a.a_bytes = (unsigned)((size_t)(((size_t)a.n_seq * a.L + 255) / 256) * 256 * a.lda * 2);
#ifdef HG_EXPERIMENTS
    static const int mode = []() { const char* e = getenv("HG_QA_MODE"); return e ? atoi(e) : 0; }();
#else
    constexpr int mode = 0;
#endif
    hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(512), QA_LDS, s, a, mode);
    return hipGetLastError();
}

}  // namespace hg
