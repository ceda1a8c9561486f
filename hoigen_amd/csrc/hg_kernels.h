// Host-callable launchers of the gfx950 kernels (internal; the public ABI is include/hoigen_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hg_common.h"

namespace hg {

// ---- GEMM:  D[m][n] = sum_k A[m][k] * W[n][k]  (fp16 operands, fp32 accumulate) ---------------
enum Epilogue : int {
    EPI_BIAS_F16 = 0,        // out fp16 [M,ldc] = acc + bias
    EPI_BIAS_QGELU_F16 = 1,  // out fp16 = quickgelu(acc + bias)        (clipnet/model.py:162-164)
    EPI_BIAS_RELU_F16 = 2,   // out fp16 = relu(acc + bias)
    EPI_BIAS_RESID_F32 = 3,  // out fp32 += acc + bias   (residual stream, in place)
    EPI_BIAS_F32 = 4,        // out fp32 = acc + bias (bias may be null)
    EPI_PATCH_F32 = 5,       // patch embedding: row m=(b,t) -> out row b*L+1+t (+ pos[1+t] when GemmArgs::pos is given)
    EPI_BIAS_RELU_F32 = 6,   // out fp32 = relu(acc + bias)
    EPI_SCALE_RESID_F32 = 7, // out fp32 += (acc + bias) * pos[n]   (adapter up_proj * scale, residual)
    // LayerNorm folded into the GEMM (ring kernels only): A = raw fp16 copy of the residual rows, W = gamma-folded
    // weight, cs[n] = sum_k W'[n][k], bias' = bias + W beta, mr[m] = (mean, rstd) of row m:
    //     out = rstd[m] * (acc - mean[m] * cs[n]) + bias'[n]
    EPI_LN_BIAS_F16 = 8,
    EPI_LN_BIAS_QGELU_F16 = 9,
    // residual update that also emits what the next folded-LN GEMM needs (ring2 kernel only):
    //     out fp32 += acc + bias;  out2 fp16 = the updated rows;  stats[m][tile_n] = (sum, sum of squares) over the tile's columns
    EPI_RESID_LN_F32 = 10,
    // A = centred fp16 copy of the stream (x - mu[m]): out fp32 = relu(acc + mu[m] * cs[n] + bias[n]) = relu(W x + b)
    // (adapter down_proj on the copy the residual GEMMs emit; simple kernel only)
    EPI_MU_BIAS_RELU_F32 = 11,
    // EPI_RESID_LN_F32 with the update scaled per column: out fp32 += (acc + bias) * pos[n]; fp16 copy + statistics
    // (adapter up_proj: LayerNorm folding stays on behind the adapter; duo kernel only)
    EPI_SCALE_RESID_LN_F32 = 12
};

// Low half of a residual stream held as centre + hi + lo (GemmArgs::hl): 1 = bf8 (e5m2: the top byte of the fp16 remainder, rounded to
// nearest even by v_cvt_pk_bf8_f32; 13-14 bits of x - centre, 6 bytes per element through a residual epilogue), 0 = fp16 (22 bits,
// 8 bytes).  One of the two is built; hg_get_option("stream_lo_bits") reports which.
#ifndef HG_LO8
#define HG_LO8 1
#endif
// The bf8 low half holds the remainder TIMES 2^10: the remainder of an fp16 rounding is <= 2^-11 of the element, so unscaled it sits
// 11 binades below hi and - e5m2 has fp16's exponent range - flushes to zero once the element is below ~0.25 (a stream whose rows
// spread by 0.02 would travel as fp16 alone); scaled it is normal wherever |x - centre| >= 2.4e-4 (hi: 6.1e-5) and cannot overflow
// (<= half of |hi|).  One packed multiply on the way out, the add on the way in becomes a packed fma.
#ifndef HG_LO_SCALE
#define HG_LO_SCALE 1024.0f
#endif

struct GemmArgs {
    const half_t* A;   // [M, lda] (K contiguous)
    const half_t* W;   // [N, K]
    const float* bias; // [N] or nullptr
    void* out;
    const float* pos;  // EPI_PATCH: positional embedding [L, N]
    int lda, ldc;
    int M, N, K;
    int G, L;          // EPI_PATCH: patches per image, tokens per image
    const float* cs = nullptr;   // EPI_LN_*: [N] column sums of the folded weight
    const float* mr = nullptr;   // EPI_LN_*: [M][2] (row mean minus the centre of the fp16 copy, rstd)
    const float* mu = nullptr;   // EPI_RESID_LN: [M] centre subtracted from the fp16 copy (the row's previous mean)
    half_t* out2 = nullptr;      // EPI_RESID_LN: fp16 copy of the updated rows, leading dimension ld2 (0 = ldc)
    int ld2 = 0;                 //   a stride other than ldc exists in gemm_ring2 only (the dispatcher routes it there)
    float* stats = nullptr;      // EPI_RESID_LN: [M][stats_ld][2] partial (sum, sumsq) per row and wave column group
    int stats_ld = 0;            //   = 4 * N / 256
    // EPI_BIAS_F32 / EPI_BIAS_RELU_F32 with two destinations: columns >= n_split go to out_hi[m * ldc + (n - n_split)]
    // (the stacked mean | log_var GEMM of the VAE encoder writes its two halves straight into the caller's tensors)
    void* out_hi = nullptr;
    int n_split = 0;                     // multiple of 16
    unsigned long long* dbg = nullptr;   // diagnostics: s_memtime totals (HG_STAMPS build), normally null
    // EPI_RESID_LN with the residual stream held as TWO fp16 halves instead of fp32 (gemm_ring2 only; DESIGN.md 4 "hi / lo"):
    //   x = centre + hi + lo,   hi = fp16(x - centre) = the centred copy `out2` the next GEMM reads anyway,
    //   lo = fp16((x - centre) - hi) in `lo` (tile-fragment order: only this kernel family reads it, 16 B per lane, whole lines)
    // hl: 0 fp32 in / fp32 out (`out`), 1 fp32 in / hi+lo out, 2 hi+lo in / hi+lo out, 3 hi+lo in / fp32 out (+ copy as always).
    // muc [M] = the centre the hi / lo being READ were written with (finalize_stats' muc); mu = the centre to write with.
    int hl = 0;
    half_t* lo = nullptr;
    const float* muc = nullptr;
    // EPI_RESID_LN_F32 (gemm_ring2): gamma [N] = the NEXT LayerNorm's weight, multiplied into the copy the next GEMM reads -
    // fp16((x - mu) * gamma) - so that GEMM runs on the layer's own fp16 weights with cs[n] = sum_k gamma[k] W[n][k] (fold_ln's csg).
    // hl 0 / 3: out2 is that scaled copy; hl 1 / 2: out2 stays the stream's unscaled hi half and the scaled copy goes to out3 (row stride ld3)
    const float* gamma = nullptr;
    half_t* out3 = nullptr;
    int ld3 = 0;
};
// bytes of the `lo` buffer for M rows x N columns (whole 128 x 256 tiles)
inline size_t gemm_lo_bytes(int M, int N) { return (size_t)((M + 127) / 128) * (N / 256) * 65536; }

// Requirements: N % 128 == 0, K % 64 == 0, A readable for rows < M, 16-byte aligned rows.
// A must be allocated with its row count padded to a multiple of 256 (the ring kernel's DMA reads whole
// tiles; rows >= M are never stored).
hipError_t launch_gemm(int epi, const GemmArgs& a, hipStream_t s);
double gemm_flops(const GemmArgs& a);
// persistent ring kernel (hg_gemm_ring.hip) and the simple 128x128 kernel (hg_gemm.hip)
bool gemm_ring_ok(const GemmArgs& a);
hipError_t launch_gemm_ring(int epi, const GemmArgs& a, hipStream_t s);
bool gemm_ring2_ok(const GemmArgs& a);
// LayerNorm-folding epilogues exist only in the ring kernels: EPI_LN_* in gemm_ring (bias + column sums must fit
// behind the ring), EPI_RESID_LN_F32 in gemm_ring2
bool gemm_ln_ok(int epi, const GemmArgs& a);
hipError_t launch_gemm_ring2(int epi, const GemmArgs& a, hipStream_t s);
hipError_t launch_gemm_simple(int epi, const GemmArgs& a, hipStream_t s);
// c_fc -> QuickGELU -> c_proj of a block as ONE persistent launch (hg_mlp_pair.hip): fc = the EPI_LN_BIAS_QGELU_F16 GEMM, proj = the
// EPI_RESID_LN_F32 GEMM that reads fc.out; ready [mlp_pair_ready_words(M)] u32 zeroed on the stream before the launch; err: host-mapped
// word set when a hand-off wait times out; ch: 256-row panels of an XCD per chunk of the c_fc tile order; fc_slots: workgroups per XCD
// that run c_fc tiles (of n_cu / 8)
bool mlp_pair_ok(const GemmArgs& fc, const GemmArgs& proj, int n_cu);
size_t mlp_pair_ready_words(int M);
// fin_mr (optional): the launch also does finalize_stats' work for the rows c_proj updates - mr / mu / muc / range_flag as in
// launch_finalize_stats (mu must be proj.mu) - in its tail; null: the caller launches finalize_stats
hipError_t launch_mlp_pair(const GemmArgs& fc, const GemmArgs& proj, unsigned* ready, int* err, int ch, int fc_slots, int n_cu, hipStream_t s,
                           float* fin_mr = nullptr, float* fin_mu = nullptr, float* fin_muc = nullptr, int* range_flag = nullptr,
                           int grid_short = 0);      // grid_short (fault injection: hg_api.hip option mlp_pair_fault): workgroups NOT launched
// two 4-wave workgroups per CU, 128x256 tiles, free-running (hg_gemm_duo.hip): residual GEMMs and fp16/fp32 outputs
bool gemm_duo_ok(int epi, const GemmArgs& a);
hipError_t launch_gemm_duo(int epi, const GemmArgs& a, hipStream_t s);

// ---- attention: softmax(Q K^T / sqrt(64) [+causal]) V, head_dim 64 --------------------------
// qkv fp16 [n_seq*L, 3*D] rows = tokens (q|k|v column blocks, head h = 64h..64h+63);
// out fp16 [n_seq*L, D].  L <= 224.
// ldo: row stride of out in halfs (0 = heads * 64)
// pack: L <= 32 (one wave per item) runs four items per workgroup (dispatch-bound otherwise; bit-identical)
hipError_t launch_attention(const half_t* qkv, half_t* out, int n_seq, int L, int heads, bool causal,
                            hipStream_t s, int ldo = 0, bool pack = true);
// attention of ONE query row per sequence (row sel[seq], row 0 when sel is null; sel[seq] < L): K and V from
// qkv [n_seq*L, 3*heads*64], the queries from q0 [n_seq, heads*64] (dense), out [n_seq, heads*64] (dense)
hipError_t launch_attention_row0(const half_t* qkv, const half_t* q0, const int32_t* sel, half_t* out, int n_seq, int L,
                                 int heads, bool causal, hipStream_t s);

// ---- fused in_proj + attention (hg_qkv_attn.hip): out = SDPA(LN-folded in_proj(x16)) with q, k, v kept in LDS ----------------
// x16 [n_seq * L, lda]: centred fp16 copy of the stream; wp / bcs: the LayerNorm-folded in_proj weight, bias' and column sums
// packed by launch_pack_qkv; mr [n_seq * L][2] = (row mean minus the copy's centre, rstd); out fp16 [n_seq * L, ldo].
struct QkvAttnArgs {
    const half_t* x16 = nullptr;
    int lda = 0;
    const half_t* wp = nullptr;
    const float* bcs = nullptr;
    const float* mr = nullptr;
    half_t* out = nullptr;
    int ldo = 0;                 // 0 = D
    int n_seq = 0, L = 0, D = 0, heads = 0;
    int K = 0;                   // row length of x16 / of W that is summed over (0 = D; D + 64 with the adapter's e columns behind x16)
    unsigned a_bytes = 0;        // bytes readable behind x16 (0: rows padded to a multiple of 256)
    int gsz = 0;                 // head pairs per XCD group (0 = all: the pairs of a sequence side by side)
    unsigned long long* dbg = nullptr;   // diagnostics: s_memtime totals per wave (HG_STAMPS build), normally null
};
// 192 < L <= 208, D = 64 * heads, heads even, D / 64 a multiple of 3 (ViT-B/16: L = 197, D = 768); K / 64 = 3 m or 3 m + 1
bool qkv_attn_ok(int n_seq, int L, int D, int heads, int lda, int K = 0);
// speed only: is the last round of (sequence, head pair) items filled well enough (n_cu <= 0: 256)
bool qkv_attn_pays(int n_seq, int heads, int n_cu);
// W [3D, K] fp16 (LayerNorm-folded), bias / cs [3D] -> Wp [3D * K] fp16 in fragment order, bcs [heads / 2][768] fp32 (null: not written)
hipError_t launch_pack_qkv(const half_t* W, const float* bias, const float* cs, half_t* Wp, float* bcs, int D, int heads,
                           hipStream_t s, int K = 0);
hipError_t launch_qkv_attn(const QkvAttnArgs& a, hipStream_t s);

// ---- CoOp-VAE as one kernel (hg_vae_fused.hip): Encoder -> reparameterise -> Generator with both hidden layers and z on chip --------
// (main_coop_vae.py:261-296,444-448).  wp = the weights packed by launch_pack_vae into the fragment stream the kernel walks:
// [E0 | E1 | G] passes (encoder halves for z columns [0,256) / [256,512), generator), each vae_fused_pass_bytes(hidden) long.
struct VaeFusedArgs {
    const float* x = nullptr;        // [R, 512] fp32: Encoder input (mode 2: the Generator's z)
    const half_t* x16 = nullptr;     // mode 2 only: z as fp16 [R, 512] instead of x (what the GEMM path's reparameterisation kernel writes)
    const float* eps = nullptr;      // [R, 512] fp32 (modes 0, 1)
    float* mean = nullptr;           // [R, 512] fp32 outputs; mean / logvar / z may be null
    float* logvar = nullptr;
    float* z = nullptr;
    float* bias = nullptr;           // Generator output (modes 0, 2)
    const half_t* wp = nullptr;
    const float* b0e = nullptr;      // Encoder.net.0.bias [eh]
    const float* bml = nullptr;      // mean.bias | log_var.bias [1024]
    const float* b0g = nullptr;      // Generator.net.0.bias [gh]
    const float* b2g = nullptr;      // Generator.net.2.bias [512]
    half_t* zpark = nullptr;         // scratch: vae_fused_park_bytes(R) (modes 0, 1)
    int R = 0, eh = 0, gh = 0;
    int mode = 0;                    // 0 Encoder + Generator, 1 Encoder only, 2 Generator only
    bool has_enc = true;             // wp starts with the two encoder passes (mode 2 skips them)
};
bool vae_fused_ok(int dim, int eh, int gh);          // dim == 512, hidden widths multiples of 32 up to 4096
size_t vae_fused_pass_bytes(int hidden);
int vae_fused_rows_per_item();                        // 128
inline size_t vae_fused_park_bytes(int R) { return (size_t)((R + 127) / 128) * 4 * 16 * 1024; }
hipError_t launch_pack_vae(const half_t* e_w0, const half_t* e_wml, int eh, const half_t* g_w0, const half_t* g_w2, int gh, half_t* wp,
                           hipStream_t s);
hipError_t launch_vae_fused(const VaeFusedArgs& a, hipStream_t s);

// ---- elementwise / row kernels ---------------------------------------------------------------
// LayerNorm over rows of fp32 x (eps 1e-5, biased variance; clipnet/model.py:153-159).
// Input row for output row r:  gather ? r*rows_per_seq + gather[r] : r*in_row_stride_rows.
hipError_t launch_layernorm_f16(const float* x, const float* w, const float* b, half_t* out, int M, int D,
                                const int32_t* gather, int rows_per_seq, int in_row_mul, hipStream_t s);
// pos / cls / L (ln_pre of the vision tower): row r is first replaced by (r % L == 0 ? cls : x[r]) + pos[r % L]
hipError_t launch_layernorm_f32(const float* x, const float* w, const float* b, float* out, int M, int D,
                                hipStream_t s, const float* pos = nullptr, const float* cls = nullptr, int L = 1);
// NCHW fp32 crops -> patch matrix fp16 [B*g*g, 3*p*p] (token t = g*row+col, k = c*p*p+ky*p+kx).
hipError_t launch_im2col(const float* x, half_t* out, int B, int R, int p, hipStream_t s);
// x[r][:] = table[ids[r/L*ld_ids + r%L]][:] + pos[r%L][:]   (text: token_embedding + positional)
hipError_t launch_embed_tokens(const int32_t* ids, int ld_ids, const float* table, const float* pos, float* x,
                               int T, int L, int D, int vocab, hipStream_t s);
// x[r][:] = prompts[(r/L)*Lfull + r%L][:] + pos[r%L][:]
hipError_t launch_add_pos(const float* prompts, int Lfull, const float* pos, float* x, int R, int L, int D,
                          hipStream_t s);
hipError_t launch_gather_rows(const int32_t* ids, const float* table, float* out, int n, int D, int vocab,
                              hipStream_t s);
hipError_t launch_eot_argmax(const int32_t* ids, int T, int L, int32_t* eot, int32_t* max_eot, hipStream_t s);
// out[i] = min(max(in[i], 0), Leff - 1); *flag = 1 (host-mapped, sticky) and *flag_dev = 1 (device memory, per call) if any
// in[i] >= Leff.  in may equal out.
hipError_t launch_clamp_eot(const int32_t* in, int n, int Leff, int32_t* out, int32_t* flag, hipStream_t s, int32_t* flag_dev = nullptr);
// out[0..n) = NaN if *flag != 0 (flag: DEVICE memory, set earlier on the same stream)
hipError_t launch_poison_if_flag(float* out, size_t n, const int32_t* flag, hipStream_t s);
hipError_t launch_f32_to_f16(const float* in, half_t* out, size_t n, hipStream_t s);
hipError_t launch_f16_to_f32(const half_t* in, float* out, size_t n, hipStream_t s);
// out[c][r] = in[r][c]  (fp32 or fp16 in -> fp16 out), used for `proj` / `text_projection` [in,out]
hipError_t launch_transpose_to_f16(const void* in, int in_dtype, half_t* out, int rows, int cols, hipStream_t s);
hipError_t launch_l2_normalize(const float* x, float* out, int R, int D, hipStream_t s);
hipError_t launch_assemble_prompts(const float* prefix, const float* suffix, const float* ctx, const float* bias,
                                   const int32_t* target, int R, int C, int L, int n_ctx, int D, float* prompts,
                                   hipStream_t s);
// z = exp(0.5 * logvar) * eps + mean  (main_coop_vae.py:445-447) from the mean / logvar planes [R, D]: z fp32 (optional)
// and z16 fp16 [R, ld16] (GEMM operand).  D % 4 == 0, 16-byte accesses.
hipError_t launch_reparam(const float* mean, const float* logvar, const float* eps, int R, int D, float* z, half_t* z16,
                          int ld16, hipStream_t s);
hipError_t launch_vae_loss(const float* recon, const float* x, const float* mean, const float* logvar, int R,
                           int D, float* loss, hipStream_t s);
// tokens [B*L, E] fp32 -> global [B,E] (token 0) and local [B,E,g,g] (tokens 1..), NCHW
hipError_t launch_split_global_local(const float* tok, float* glob, float* local, int B, int L, int E,
                                     hipStream_t s);
hipError_t launch_copy_rows(const float* x, float* out, int B, int row_stride, int D, hipStream_t s,
                            const int32_t* gather = nullptr);   // row b*row_stride (+ gather[b])
// the same from a stream held as centre + hi + lo (GemmArgs::hl): hi [*, D] row-major, lo in gemm_ring2's tile-fragment order
hipError_t launch_copy_rows_hilo(const half_t* hi, const half_t* lo, const float* muc, float* out, int B, int row_stride, int D,
                                 hipStream_t s);
// first N columns of a [R, ld] fp32 matrix -> dense [R, N]
hipError_t launch_copy_cols(const float* x, int ld, float* out, int R, int N, hipStream_t s);
// ---- LayerNorm folding support (DESIGN.md §4 "LayerNorm folded into the GEMMs")
// W16 [N,K], gamma/beta [K], bias [N]  ->  Wf16 = fp16(W * gamma), cs[n] = sum_k float(Wf16[n][k]), bf[n] = bias[n] + sum_k W[n][k] * beta[k]
hipError_t launch_fold_ln(const half_t* w16, const float* gamma, const float* beta, const float* bias, half_t* wf16,
                          float* cs, float* bf, int N, int K, hipStream_t s, float* csg = nullptr);
// x fp32 [M,D] -> centred fp16 copy x16 = fp16(x - mean), mu[m] = mean, mr[m] = (0, rstd) (eps 1e-5, biased variance)
hipError_t launch_rowstats_cast(const float* x, half_t* x16, float* mr, float* mu, int M, int D, hipStream_t s,
                                float* muc = nullptr, const float* gamma = nullptr);   // muc (optional): centre of the copy as well (= mu)
// x = LayerNorm(x; w, b) in place (fp32) followed by rowstats_cast of the result, in one pass (ln_pre of the vision tower)
hipError_t launch_layernorm_rowstats(float* x, const float* w, const float* b, half_t* x16, float* mr, float* mu, float* muc,
                                     int M, int D, hipStream_t s, const float* pos = nullptr, const float* cls = nullptr,
                                     int L = 1, int ld16 = 0);      // ld16: row stride of x16 in halfs (0 = D)
// stats [M][nt][2]: per column group of gw columns (sum, sum of squared deviations from the group mean)
// -> mr [M][2] = (mean - mu[m], rstd) over the nt * gw columns, eps 1e-5, then mu[m] = mean (the centre the next
// residual GEMM subtracts from its fp16 copy)
// muc (optional) receives the centre the CURRENT fp16 copy was written with (mu before this call)
// centred: the statistics are those of ALREADY CENTRED values (EPI_X16_SCALE_LN): mr = (mean, rstd), mu stays
// range_flag (optional, host-mapped): set to 1 when a row's reach from its centre mu[m], bounded through the statistics, exceeds fp16's 65 504
hipError_t launch_finalize_stats(const float* stats, float* mr, float* mu, int M, int nt, int gw, hipStream_t s,
                                 float* muc = nullptr, bool centred = false, int* range_flag = nullptr);
#define HG_PRE_HDR 24   // header words per box in the pre-processing table (layout: hg_preproc.hip)
// ---- crop pre-processing (hg_preproc.hip): head = per-box headers written by the host, tab receives the weight
// tables at word offsets tab_off[box]; tmp = uint8 scratch for the horizontal pass; out fp32 [n,3,n_px,n_px]; out_u8 (nullable) uint8
// [n,n_px,n_px,3] before normalisation
hipError_t launch_preprocess(const uint8_t* img, int H, int W, const int32_t* head, int32_t* tab,
                             const int32_t* tab_off, int n, int n_px, int max_rows, uint8_t* tmp, float* out,
                             uint8_t* out_u8, hipStream_t s, bool imagenet_norm = false);

// RoI-align (torchvision semantics, aligned=True, sampling_ratio=-1) of feat [C,H,W] for boxes [n,4] (device):
// out_pooled [n,C,P,P] and/or out_mean [n,C] (= .flatten(2).mean(-1)); either may be null
hipError_t launch_roi_align(const float* feat, int C, int H, int W, const float* boxes, int n, float spatial_scale,
                            int P, float* out_pooled, float* out_mean, hipStream_t s);

// ---- adapter (variant C) --------------------------------------------------------------------
struct AdapterDev {      // device pointers, all fp32 except the two MFMA operands
    const half_t* down_w;   // [d, D] fp16
    const float* down_b;    // [d]
    const half_t* up_w;     // [D, d] fp16
    const float* up_b;      // [D]
    const float* scale;     // [D]
    // decoder layers [0] = mhsa_layers.0 (prior memory), [1] = mhsa (self memory); fp32, weights
    // transposed to [in][out]:  0 WqT 1 WkT 2 WvT 3 bq 4 bk 5 bv 6 WoT 7 bo 8 {norm2.w,norm2.b,norm3.w,
    // norm3.b} 9 W1T [d][2d] 10 b1 11 {W2T [2d][d], b2}
    const float* dl[2][12];
    // fp16 [out][in] copies for the MFMA decoder: 0 Wq 1 Wk 2 Wv [64,64] (rows of in_proj_weight), 3 Wo [64,64],
    // 4 W1 [128,64], 5 W2 [64,128]; null -> the fp32 VALU kernel runs
    const half_t* w16[2][6];
};
// down32 [M,128] fp32 (cols 0..63 = relu(down_proj(x))) -> out16 [M,64] fp16 (decoder layer output);
// kv: scratch [B*Nmem, 2, 64] fp32 with Nmem = N (prior given) or L (prior == nullptr).
// chain32 (optional, adapter_num_layers > 1): write the layer's output as fp32 into this [M,128] buffer (may be down32
// itself) instead of out16, for the next decoder layer of the chain
// ld16: row stride of out16 in halfs (64 = dense; D + 64 when the rows are the right-hand columns of the K-concatenated
// out-proj operand [att | d])
// fold (MFMA path, last layer of a chain only): the layer writes e = [z_0 .. z_62, 1] (norm3 without its affine part) instead
// of d and replaces mr[row] = (mean_x - c, rstd_x) by the LayerNorm statistics of x + a, a = Q e (adapter_q_kernel); it reads
// w' = x16 Q from columns 64..127 of down32
struct AdapterFoldDev {
    const half_t* g16 = nullptr;   // [64][64] fp16 Q^T Q
    const float* qm = nullptr;     // [64] column sums of Q
    float* mr = nullptr;           // [M][2], updated in place; null = off
    float inv_D = 0.f;
    half_t* e2 = nullptr;          // a second home of e (row stride ld16): the out-proj operand buffer when it is not the in_proj one
};
// dn (MFMA path, first layer of a chain): down_proj runs inside the kernel - [relu(down) | x16 Q] = [x16 + muc] w^T + b with
// w [128, K] fp16 (AdapterW::Fold::down2), b / cs [128]; down32 is then only written (chain32: the fp32 hand-over to the next
// layer, x16 Q in its columns 64..127) or not touched at all
struct AdapterDownDev {
    const half_t* x16 = nullptr;   // [M, ldx] centred fp16 copy of the stream
    int ldx = 0, K = 0;
    const half_t* w = nullptr;
    const float* b = nullptr;
    const float* cs = nullptr;
    const float* muc = nullptr;    // [M] centre of the copy
};
hipError_t launch_adapter_decoder(const float* down32, const AdapterDev& ad, const float* priors,
                                  const uint8_t* mask, int B, int L, int N, float* kv, half_t* out16,
                                  hipStream_t s, float* chain32 = nullptr, int ld16 = 64, const AdapterFoldDev* fold = nullptr,
                                  const AdapterDownDev* dn = nullptr);
bool adapter_decoder_mfma_ok(const AdapterDev& ad, bool priors, int L, int N);
bool adapter_decoder_fused_down_ok(const AdapterDev& ad, bool priors, int L, int N);
// weight-load time: Q (scratch q32 [D][64]) and the operands that carry it: down2 [128, D], wk_out [D, D+64] = [w_out | Q],
// wq_cat [3D, D+64] = [wf_qkv | wf_qkv Q], qm [64], g16 [64][64]; norms = the last decoder layer's {norm2.w, norm2.b, norm3.w, norm3.b}
hipError_t launch_adapter_fold(const half_t* up_w, const float* up_b, const float* scale, const float* norms,
                               const half_t* down_w, const half_t* w_out, const half_t* wf_qkv, int D, float* q32,
                               half_t* down2, half_t* wk_out, half_t* wq_cat, float* qm, half_t* g16, hipStream_t s);

}  // namespace hg
