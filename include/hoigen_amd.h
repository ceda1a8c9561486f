/*
 * hoigen_amd — C ABI of the MI355X (gfx950) hot path of HOIGen.
 *
 * The reference (soberguo/HOIGen) has no FFI: its boundary is Python duck-typing on nn.Module
 * attributes.  This header is the boundary a binding for that path would target;
 * `hoigen_amd/_lib.py` is the ctypes binding and `hoigen_amd/{clip,model,vae}.py` present the
 * reference's own Python signatures on top of it.  Every entry point names the reference
 * interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - All data pointers are DEVICE pointers to contiguous row-major buffers owned by the caller.
 *  - Calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *  - Return value: 0 on success, negative hg_status on failure; text via hg_last_error().
 *  - One context per device per process; a context is not re-entrant (the reference is
 *    single-threaded per process, one process per GPU: main_tip_finetune.py:1205-1208).
 *  - The library allocates device memory only in hg_create / hg_load_* / on workspace growth
 *    (monotonic); steady-state calls do not allocate or synchronise (hipGraph-capturable).
 *  - Arithmetic: fp16 MFMA operands, fp32 accumulation; residual stream, LayerNorm statistics and
 *    softmax in fp32.  Inference only (no autograd).
 */
#ifndef HOIGEN_AMD_H
#define HOIGEN_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hg_ctx hg_ctx;

typedef enum {
    HG_OK = 0,
    HG_ERR_INVALID = -1,     /* bad argument / unsupported shape            */
    HG_ERR_HIP = -2,         /* a HIP runtime call failed                   */
    HG_ERR_NOT_LOADED = -3,  /* weights for this entry point not loaded yet */
    HG_ERR_OOM = -4
} hg_status;

typedef enum { HG_F32 = 0, HG_F16 = 1 } hg_dtype;

/* A weight tensor handed to hg_load_*: device pointer + element type.  ptr == NULL means absent. */
typedef struct {
    const void* ptr;
    int32_t dtype; /* hg_dtype */
} hg_tensor;

/* ResidualAttentionBlock parameters — clipnet/model.py:167-188; state-dict keys
 * `{prefix}.resblocks.{i}.{attn.in_proj_weight,...}`.  Linear weights are [out,in]. */
typedef struct {
    hg_tensor in_proj_weight;  /* [3D, D]  rows q|k|v */
    hg_tensor in_proj_bias;    /* [3D]                */
    hg_tensor out_proj_weight; /* [D, D]              */
    hg_tensor out_proj_bias;   /* [D]                 */
    hg_tensor ln_1_weight, ln_1_bias; /* [D] */
    hg_tensor c_fc_weight;     /* [4D, D] */
    hg_tensor c_fc_bias;       /* [4D]    */
    hg_tensor c_proj_weight;   /* [D, 4D] */
    hg_tensor c_proj_bias;     /* [D]     */
    hg_tensor ln_2_weight, ln_2_bias; /* [D] */
} hg_block_weights;

/* TransformerDecoderLayer(64, nhead=2, ffn=128) as used by the adapter —
 * CLIP_models_adapter_prior2.py:27-72 (forward_post; norm1 exists but is unused). */
typedef struct {
    hg_tensor attn_in_proj_weight;  /* [3d, d] */
    hg_tensor attn_in_proj_bias;    /* [3d]    */
    hg_tensor attn_out_proj_weight; /* [d, d]  */
    hg_tensor attn_out_proj_bias;   /* [d]     */
    hg_tensor linear1_weight, linear1_bias; /* [2d, d], [2d] */
    hg_tensor linear2_weight, linear2_bias; /* [d, 2d], [d]  */
    hg_tensor norm2_weight, norm2_bias;     /* [d] */
    hg_tensor norm3_weight, norm3_bias;     /* [d] */
} hg_decoder_layer_weights;

/* Adapter — CLIP_models_adapter_prior2.py:142-203; keys `...resblocks.{i}.adaptermlp.*`. */
typedef struct {
    int32_t present;                 /* 0: this block has no adapter (adapter_pos, :958-967) */
    int32_t bottleneck;              /* 64 */
    hg_tensor scale;                 /* [D]   */
    hg_tensor down_proj_weight, down_proj_bias; /* [d, D], [d] */
    hg_tensor up_proj_weight, up_proj_bias;     /* [D, d], [D] */
    hg_decoder_layer_weights prior_layer;       /* mhsa_layers.0 (memory = prior tokens) */
    hg_decoder_layer_weights self_layer;        /* mhsa          (memory = down itself)  */
    /* adapter_num_layers > 1 (CLIP_models_adapter_prior2.py:150,179,190-195): mhsa_layers.1 .. N-1, applied one after
     * the other on the prior path; NULL / 0 for the default single layer */
    int32_t n_extra_prior_layers;
    const hg_decoder_layer_weights* extra_prior_layers;
} hg_adapter_weights;

/* VisionTransformer — clipnet/model.py:202-236 (variant A) and
 * CLIP_models_adapter_prior2.py:471-506 (variant C). */
typedef struct {
    int32_t width, layers, heads, patch_size, input_resolution, output_dim;
    hg_tensor conv1_weight;          /* [D, 3, p, p]     */
    hg_tensor class_embedding;       /* [D]              */
    hg_tensor positional_embedding;  /* [g*g+1, D]       */
    hg_tensor ln_pre_weight, ln_pre_bias, ln_post_weight, ln_post_bias; /* [D] */
    hg_tensor proj;                  /* [D, E]  (x @ proj) */
    const hg_block_weights* blocks;      /* [layers] */
    const hg_adapter_weights* adapters;  /* NULL or [layers] */
} hg_vit_weights;

/* Text tower of CLIP — clipnet/model.py:282-292,339-352. */
typedef struct {
    int32_t width, layers, heads, context_length, vocab_size, output_dim;
    hg_tensor token_embedding;       /* [V, D]  */
    hg_tensor positional_embedding;  /* [L, D]  */
    hg_tensor ln_final_weight, ln_final_bias; /* [D] */
    hg_tensor text_projection;       /* [D, E]  (x @ text_projection) */
    const hg_block_weights* blocks;  /* [layers] */
} hg_text_weights;

/* CoOp-VAE — main_coop_vae.py:261-296.  Any of the two halves may be absent (ptr NULL). */
typedef struct {
    int32_t dim, enc_hidden, gen_hidden;       /* 512, 2048, 4096 */
    hg_tensor enc_w0, enc_b0;                  /* Encoder.net.0     [2048,512],[2048] */
    hg_tensor enc_mean_w, enc_mean_b;          /* Encoder.mean      [512,2048],[512]  */
    hg_tensor enc_logvar_w, enc_logvar_b;      /* Encoder.log_var   [512,2048],[512]  */
    hg_tensor gen_w0, gen_b0;                  /* Generator.net.0   [4096,512],[4096] */
    hg_tensor gen_w2, gen_b2;                  /* Generator.net.2   [512,4096],[512]  */
} hg_vae_weights;

/* mlp_net(512,512,512) — finetune_ship.py:302-314 / main_tip_finetune.py:313-324. */
typedef struct {
    int32_t in_dim, hidden_dim, out_dim;
    hg_tensor w0, b0, w2, b2, w4, b4;
} hg_mlp_weights;

#define HG_MAX_SLOTS 16 /* independent VAE / mlp_net weight sets per context: three branches (hoi, human, object) x Encoder, Generator, VAE + spares */

/* ---- lifecycle ------------------------------------------------------------------------- */
hg_ctx* hg_create(int device);                 /* NULL on failure (no HIP device)               */
void hg_destroy(hg_ctx*);
const char* hg_last_error(hg_ctx*);            /* valid until the next call on the context      */
const char* hg_version(void);

/* ---- weights (replace nn.Module.load_state_dict / build_model: clipnet/model.py:395-432,
 *      CLIP_models_adapter_prior2.py:934-984).  Weights are copied/converted; the caller may free
 *      its tensors afterwards.  Loading again replaces the previous set. -------------------------- */
int hg_load_vit(hg_ctx*, const hg_vit_weights*);
int hg_load_text(hg_ctx*, const hg_text_weights*);
int hg_load_vae(hg_ctx*, int slot, const hg_vae_weights*);
int hg_load_mlp(hg_ctx*, int slot, const hg_mlp_weights*);
/* Refresh only the adapter tensors of an already-loaded ViT (they are the trainable part:
 * main_tip_finetune.py:955-962). */
int hg_update_adapters(hg_ctx*, const hg_adapter_weights* adapters, int layers);

/* ---- image tower --------------------------------------------------------------------------- */
/* CLIP.encode_image / VisionTransformer.forward, variant A (clipnet/model.py:219-236,336-337):
 * x [B,3,R,R] fp32 NCHW -> out [B,E] fp32. */
int hg_encode_image(hg_ctx*, const float* x_nchw, int B, float* out, void* stream);
/* Variant C VisionTransformer.forward(x, prior) (CLIP_models_adapter_prior2.py:489-506).
 * priors [B,N,64] fp32 and mask [B,N] uint8 (1 = padded key) or both NULL / N == 0 for prior=None.
 * out_global [B,E]; out_local [B,E,g,g] (NCHW).  Blocks without a loaded adapter behave as A. */
int hg_encode_image_prior(hg_ctx*, const float* x_nchw, const float* priors, const uint8_t* mask,
                          int B, int N, float* out_global, float* out_local_nchw, void* stream);
/* Test hook: as hg_encode_image, additionally copies the CLS row of the residual stream after
 * ln_pre and after every block into trace [(layers+1), B, D] fp32. */
int hg_encode_image_trace(hg_ctx*, const float* x_nchw, int B, float* out, float* trace, void* stream);

/* ---- text tower ---------------------------------------------------------------------------- */
/* CLIP.encode_text (clipnet/model.py:339-352): ids [T,L] int32 (L <= context_length), zero padded,
 * EOT = largest id per row -> out [T,E] fp32.  `trunc` != 0 runs the causal tower only on the first
 * max(EOT)+1 positions (identical selected outputs; SURVEY.md §5 "Long-context"). */
int hg_encode_text_ids(hg_ctx*, const int32_t* ids, int T, int L, float* out, int trunc, void* stream);
/* TextEncoder.forward(prompts, tokenized_prompts) (main_coop_vae.py:54-63): prompts [R,L,D] fp32
 * already embedded, eot_idx [R] int32 = tokenized_prompts.argmax(-1) -> out [R,E] fp32. */
int hg_encode_text_embeds(hg_ctx*, const float* prompts, const int32_t* eot_idx, int R, int L,
                          float* out, int trunc, void* stream);
/* token_embedding(ids) (clipnet/model.py:340): ids [n] int32 -> out [n,D] fp32 (exact gather). */
int hg_token_embedding(hg_ctx*, const int32_t* ids, int n, float* out, void* stream);

/* ---- CoOp-VAE ------------------------------------------------------------------------------ */
/* Encoder -> reparameterise(eps given) -> Generator (main_coop_vae.py:444-448).
 * x, eps [R,dim] fp32 -> mean, logvar, z, bias [R,dim] fp32 (any output may be NULL). */
int hg_vae_forward(hg_ctx*, int slot, const float* x, const float* eps, int R, float* mean,
                   float* logvar, float* z, float* bias, void* stream);
/* Generator.forward (main_coop_vae.py:293-296): z [R,dim] -> bias [R,dim]. */
int hg_generator(hg_ctx*, int slot, const float* z, int R, float* bias, void* stream);
/* mlp_net.forward (finetune_ship.py:312-314). */
int hg_mlp_net(hg_ctx*, int slot, const float* x, int R, float* out, void* stream);
/* PromptLearner_*.forward (main_coop_vae.py:119-128): prefix [C,1,D], suffix [C,L-1-n_ctx,D],
 * ctx [n_ctx,D], bias [R,D], target [R] int32 -> prompts [R,L,D]; all fp32. */
int hg_assemble_prompts(hg_ctx*, const float* prefix, const float* suffix, const float* ctx,
                        const float* bias, const int32_t* target, int R, int C, int L, int n_ctx,
                        int D, float* prompts, void* stream);
/* x / x.norm(dim=-1, keepdim=True) (main_coop_vae.py:438,466); in == out allowed. */
int hg_l2_normalize(hg_ctx*, const float* x, int R, int D, float* out, void* stream);
/* RoI-align over the variant-C local feature map (SURVEY.md 8f-4;
 * upt_tip_cache_model_free_finetune_distill3.py:1026-1037): torchvision.ops.roi_align(feat[None], [boxes],
 * output_size=(P,P), spatial_scale, sampling_ratio=-1, aligned=True) -> out_pooled [n,C,P,P] (nullable) and its
 * .flatten(2).mean(-1) -> out_mean [n,C] (nullable).  feat: fp32 [C,H,W] (one image of hg_encode_image_prior's
 * out_local), boxes: fp32 [n,4] (x1,y1,x2,y2) in image pixels; all device pointers. */
int hg_roi_align(hg_ctx*, const float* feat, int C, int H, int W, const float* boxes, int n, float spatial_scale,
                 int P, float* out_pooled, float* out_mean, void* stream);

/* Cache-model (Tip-adapter) logits on the embeddings (SURVEY.md 8f-3;
 * upt_tip_cache_model_free_finetune_distill3.py:1158-1170):
 *   phi = f @ weight.T + bias ; logits = (phi @ labels) / sample_lens / post_div          -> [R, C]
 * weight [S,K] (cached, L2-normalised embeddings; K % 64 == 0), bias [S] (nullable = 0), labels [S,C] multi-hot
 * (values exactly representable in fp16), sample_lens [C], post_div (2 for the HO branch, else 1).
 * With labels.ptr == NULL the slot is a plain linear map (logits_text = f @ weight.T + bias -> [R, S]). */
#define HG_MAX_CACHE_SLOTS 8
typedef struct {
    hg_tensor weight, bias, labels, sample_lens;
    int32_t S, K, C;
    float post_div;
} hg_cache_weights;
int hg_load_cache(hg_ctx*, int slot, const hg_cache_weights* w);
int hg_cache_logits(hg_ctx*, int slot, const float* feats, int R, float* out, void* stream);

/* Crop pre-processing in front of encode_image: for every box of one image
 *   image.crop(box)                       pre_images/crop_images.py:204-219 (PIL: zeros outside the image)
 *   [expand2square(crop, background)]     utils_tip_cache_and_union_finetune.py:201-212  (pad_square != 0)
 *   Resize(n_px, BICUBIC), CenterCrop(n_px), ToTensor, Normalize      clipnet/clip.py:75-82
 * with Pillow's 8-bit resampling arithmetic (bit-exact uint8).  img: uint8 [H,W,3] RGB on the device;
 * boxes_host: int32 [n][4] = (x0, y0, x1, y1) in HOST memory (PIL convention, may leave the image);
 * background: 0x00BBGGRR; out: fp32 [n,3,n_px,n_px] (device); out_u8: optional uint8 [n,n_px,n_px,3] (device),
 * the resized crops before normalisation.
 * `pad_square` is a set of flags: HG_PRE_PAD_SQUARE (expand2square first), HG_PRE_STRETCH (the detector's CLIP view,
 * IResize([n_px, n_px]): both sides are resized to n_px, no centre crop; detr/datasets/transforms_clip.py:139-171,
 * :279-288), HG_PRE_IMAGENET_NORM (Normalize with the ImageNet constants of utils_tip_cache_and_union_finetune.py:86-89
 * instead of CLIP's). */
#define HG_PRE_PAD_SQUARE 1
#define HG_PRE_STRETCH 2
#define HG_PRE_IMAGENET_NORM 4
int hg_preprocess_crops(hg_ctx*, const uint8_t* img, int H, int W, const int32_t* boxes_host, int n, int n_px,
                        int pad_square, uint32_t background, float* out, uint8_t* out_u8, void* stream);
/* vae_loss forward value (main_coop_vae.py:300-303) -> loss[1] fp32. */
int hg_vae_loss(hg_ctx*, const float* recon, const float* x, const float* mean, const float* logvar,
                int R, int D, float* loss, void* stream);

/* ---- behaviour options (no reference counterpart) ------------------------------------------
 * Per context; the environment only supplies the initial values at hg_create (variable in brackets).  Keys:
 *   "last_block_row0" [HG_LAST_BLOCK_ROW0] 1: towers without token outputs run their LAST block on the one row per sequence
 *                      that reaches the output (class / EOT token); 0: on every row, like the reference
 *   "ln_fuse"         [HG_LN_FUSE]         1: LayerNorm folded into the GEMMs (both towers, calls of at least 512 rows; the text tower's
 *                      form: "text_ln_fold"); 0: separate LayerNorm kernels everywhere
 *   "adapter_fuse"    [HG_ADAPTER_FUSE]    1: ... also around the instance adapters of variant C
 *   "adapter_fold"    [HG_ADAPTER_FOLD]    1: adapter update folded into the block's own GEMMs; 0: separate up_proj GEMM
 *   "stream_hilo"     [HG_STREAM_HILO]     1: between the LayerNorm-folded blocks of variant A, of the text tower (and of variant C when every
 *                      block's adapter is folded into its GEMMs) the residual stream is held as
 *                      centre + hi + lo (the centred fp16 copy the GEMMs read + its remainder as bf8; fp16 in a -DHG_LO8=0 build:
 *                      read-only option "stream_lo_bits" = 8 / 16); 0: as fp32 throughout
 *   "qkv_attn"        [HG_QKV_ATTN]        1: in the LayerNorm-folded blocks of the vision tower in_proj and attention run as ONE kernel
 *                      (hoigen_amd/csrc/hg_qkv_attn.hip: q, k, v stay in LDS; 192 < tokens <= 208, i.e. ViT-B/16); 0: two kernels with the
 *                      qkv matrix in HBM between them; 2: the one kernel wherever the shapes allow (1 also asks that the last round
 *                      of its (sequence, head pair) items is well filled: speed only).  Bit-identical results either way.
 *   "qkv_attn_min_seq" [HG_QKV_ATTN_MIN_SEQ] ... from this many sequences per call on (default 32)
 *   "qkv_attn_gsz"    [HG_QKV_ATTN_GSZ]    head pairs per XCD group of that kernel (0 = all six side by side; speed only)
 *   "text_ln_fold"    [HG_TEXT_LN_FOLD]    how the text tower's LayerNorms reach its GEMMs.  1 (default): folded, with the LayerNorm weight
 *                      multiplied into the fp16 ACTIVATION copy the residual GEMM hands on, so that in_proj / c_fc run on the layer's own
 *                      fp16 weights as the reference does (600 prompts x 77 tokens: 5.35 -> 5.15 ms; against the reference's outputs 6.2e-4,
 *                      worst prompt 7.5e-4); 0: separate LayerNorm kernels (6.5e-4, worst prompt 7.9e-4); 2: the weight folded into
 *                      fp16(W * gamma) as in the vision tower - the fastest (4.8 ms) and, through that second rounding of the weights, the
 *                      least close (7.6e-4, worst prompt 9.6e-4 of the 1e-3 tolerance).  Calls of fewer than 512 rows (prompts x executed
 *                      tokens) take the separate kernels whatever the setting (as the vision tower does below 512 rows): a prompt's bits
 *                      depend on which of the two paths its call takes, never on its neighbours in the call.
 *   "qkv_attn_c"      [HG_QKV_ATTN_C]      1 (default): also in the blocks whose instance adapter is folded into in_proj (variant C on the
 *                      hi / lo stream: the same kernel summing over D + 64 columns); 0: those blocks keep the two kernels.  Bit-identical.
 *   "vae_fused"       [HG_VAE_FUSED]       1: hg_vae_forward / hg_generator run Encoder -> reparameterise -> Generator as ONE kernel
 *                      (hoigen_amd/csrc/hg_vae_fused.hip: both hidden layers and z stay on chip; dim 512, hidden widths multiples of 32
 *                      up to 4096) for the leading rows that fill whole rounds of its 128-row work items over the CUs, the GEMM path for
 *                      the rest (and for calls too small to fill 70 % of one round); 2: the one kernel for every row; 0: GEMM path only.
 *                      Same fp16 operand roundings on both paths; results differ by fp32 summation order.
 *   "mlp_fused"       [HG_MLP_FUSED]       blocks of width 512 (the text tower, separate-LayerNorm path): 1 = c_fc -> QuickGELU -> c_proj ->
 *                      residual as ONE kernel (hg_vae_fused.hip mode 3: the [rows, 2048] activation stays on chip) for the leading rows that
 *                      fill whole rounds of 128-row items, 2 = every row, 0 (default: measured a tie) = the two GEMMs
 *   "mlp_pair"        [HG_MLP_PAIR]        1 (default): in the LayerNorm-folded blocks of both towers (variant A; every form of the residual stream and
 *                      of the LayerNorm weight) c_fc -> QuickGELU -> c_proj run as ONE persistent launch (hoigen_amd/csrc/hg_mlp_pair.hip: the c_fc
 *                      tiles publish per-256-row-panel ready counters, the c_proj tiles of a panel - on the same XCD by its hardware id -
 *                      wait for them; results do not depend on workgroup placement; a wait that times out makes the NEXT call return
 *                      HG_ERR_HIP; devices with 8 x 32 CUs, otherwise the two launches run; the launch's workgroups wait for each other: the library
 *                      orders such launches across the streams of one process; PROCESSES that share a GPU and both run it can hold each other up
 *                      until that bound - set 0 there); 2: as 1, and the last workgroup of a row half also
 *                      combines that half's LayerNorm partial sums (no finalize_stats launch between the blocks; measured slower in the
 *                      vision tower, a tie in the text tower); 0: two launches.  Bit-identical results in all three.
 *   "mlp_pair_chunk"  [HG_MLP_PAIR_CHUNK]  1 .. 64: 256-row panels of an XCD per chunk of that launch's c_fc tile order (default 32: the XCD's whole list
 *                      column group by column group, as in the stand-alone kernel; speed only)
 *   "mlp_pair_fc_slots" [HG_MLP_PAIR_FC_SLOTS] 1 .. 64: workgroups per XCD (of 32) that run c_fc tiles in that launch (default 32); the others start with
 *                      their c_proj tiles, i.e. sleep until the first row panels are complete (speed only)
 *   "mlp_pair_fault"  [HG_MLP_PAIR_FAULT]  test hook, default 0.  1: that launch goes out one workgroup short - a work slot nobody takes -, so that the
 *                      hand-off waits that depend on it meet their bound (~0.6 s), the results of the call are wrong and the NEXT call returns
 *                      HG_ERR_HIP (tests/test_gpu_scale.py exercises "an error, never a hang" with it, and the recovery once it is 0 again)
 *   "chunk_rows"      [HG_CHUNK_ROWS]      rows per VAE / mlp_net / cache-logits chunk (>= 256; default 32768; the last chunk of a
 *                      call absorbs a tail of up to an eighth of it)
 * Unknown keys and out-of-range values return HG_ERR_INVALID. */
int hg_set_option(hg_ctx*, const char* key, int value);
int hg_get_option(hg_ctx*, const char* key, int* value);

/* ---- introspection for bench.py (no reference counterpart) --------------------------------- */
int hg_workspace_bytes(hg_ctx*, uint64_t* bytes);
/* Live per-kernel timing: from hg_profile_begin until hg_profile_end every launch of kernel kind `kind` (or, with
 * HG_PROF_ALL, every GEMM and attention launch of the towers and the VAE) is bracketed by a hipEvent pair on the
 * stream it is launched on (an event record costs a few microseconds on the stream: profile a few steps, not all).
 * GEMM kinds are the epilogue classes: 0 bias->f16, 1 bias+QuickGELU->f16, 2 bias+ReLU->f16, 3 bias+residual f32,
 * 4 bias->f32, 5 patch embedding, 6 bias+ReLU->f32, 7 scale+residual, 8 LayerNorm-folded bias->f16 [QKV],
 * 9 LayerNorm-folded bias+QuickGELU->f16 [c_fc], 10 residual + fp16 copy + row statistics [out_proj, c_proj],
 * 11 adapter down_proj on the centred copy, 12 adapter up_proj (scaled residual + fp16 copy + statistics).  hg_profile_end synchronises and returns one record per
 * timed launch, in launch order. */
#define HG_PROF_OFF (-1)
#define HG_PROF_ALL (-2)
#define HG_PROF_ATTENTION 100 /* M = sequences, N = tokens per sequence, K = heads */
#define HG_PROF_QKV_ATTN 101  /* fused in_proj + attention: M = sequences, N = tokens per sequence, K = heads */
#define HG_PROF_VAE_FUSED 102 /* CoOp-VAE as one kernel: M = rows, N = passes per item (3 Encoder + Generator, 2 Encoder, 1 Generator), K = hidden width of the first pass */
#define HG_PROF_MLP_PAIR 103  /* c_fc -> QuickGELU -> c_proj as one launch: M = rows, N = hidden width (4 D), K = D */
typedef struct {
    int32_t kind; /* GEMM epilogue class or HG_PROF_ATTENTION */
    int32_t M, N, K;
    float ms;
} hg_prof_rec;
int hg_profile_begin(hg_ctx*, int kind, int max_launches);
int hg_profile_end(hg_ctx*, hg_prof_rec* recs, int max_recs, int32_t* n_recs);
/* Test hook for the GEMM kernels: out[M,N] fp32 = epilogue(fp16(a[M,K]) x fp16(w[N,K])^T) (for the residual
 * epilogue `out` is read-modify-written).  epi: 0 bias->f16, 1 bias+QuickGELU->f16, 2 bias+ReLU->f16,
 * 3 bias+residual(f32), 4 bias->f32, 6 bias+ReLU->f32.  kernel: 0 auto, 1 simple 128x128, 2 persistent ring. */
int hg_test_gemm(hg_ctx*, const float* a, const float* w, const float* bias, float* out, int M, int N, int K,
                 int epi, int kernel, void* stream);
/* Test hook for the LayerNorm-folded / statistics-emitting epilogues the vision tower actually runs:
 *   epi 8 / 9  (ring):  out = fp16( rstd[m] * (acc - mean[m] * cs[n]) + bias[n] ) [9: QuickGELU first], mr = [M][2] (mean, rstd)
 *   epi 10     (ring2 / duo): x[M,N] (in `out`, read-modify-written) += acc + bias; out2 = fp16(x' - mu[m]);
 *              per-row partial statistics -> finalize_stats -> mr_out [M][2] = (mean - mu[m], rstd), mu_out [M] = mean
 *   epi 12     (duo, K >= 64): as 10 with the update scaled per column: x += (acc + bias) * scale[n]
 * All pointers are device fp32; a / w are rounded to fp16 inside; out2 comes back as fp32.  kernel: 0 dispatcher, 2 ring
 * family, 3 duo.  Unused pointers may be NULL. */
int hg_test_gemm_ln(hg_ctx*, const float* a, const float* w, const float* bias, float* out, int M, int N, int K, int epi,
                    int kernel, const float* cs, const float* mr, const float* mu, const float* scale, float* out2,
                    float* mr_out, float* mu_out, void* stream);
/* Test hook for the residual stream held as centre + hi + lo: `steps` (2..16) residual updates
 * x += a W^T + bias in a row through gemm_ring2 + finalize_stats, the stream between them as centre + hi + lo (hilo = 1: the
 * first update reads fp32 x, the last writes fp32 x) or as fp32 throughout (hilo = 0).  x [M,N] and mu [M] (centre of the
 * first copy in, last mean out) are read-modify-written; out2 = the last centred copy (fp32), mr_out [M][2]; both may be NULL. */
int hg_test_gemm_hilo(hg_ctx*, const float* a, const float* w, const float* bias, float* x, int M, int N, int K, int steps,
                      int hilo, float* mu, float* out2, float* mr_out, void* stream);
/* Test hook for the attention kernels (clipnet/model.py:171,181-183: the SDPA inside nn.MultiheadAttention, head_dim
 * 64): qkv [n_seq*L, 3*heads*64] fp32 on the device (rounded to fp16 inside).  q0 == NULL: full attention, out
 * [n_seq*L, heads*64].  q0 != NULL: [n_seq, heads*64] queries of ONE row per sequence (row sel[seq], device int32, or
 * row 0 when sel is NULL - the row index only matters for the causal mask), out [n_seq, heads*64]. */
int hg_test_attention(hg_ctx*, const float* qkv, const float* q0, const int32_t* sel, int n_seq, int L, int heads,
                      int causal, float* out, void* stream);

/* Test hook for the fused in_proj + attention kernel (hoigen_amd/csrc/hg_qkv_attn.hip; the reference ops it replaces:
 * clipnet/model.py:171,181-183).  a [n_seq*L, D] fp32 (rounded to fp16 inside: the centred copy of the stream), w [3D, D] the
 * LayerNorm-folded in_proj weight, bias / cs [3D], mr [n_seq*L, 2] = (mean - centre, rstd), D = 64 * heads; out [n_seq*L, D]
 * fp32 = the attention output.  fused bit 0 set: the one kernel (192 < L <= 208, heads even, D / 64 a multiple of 3);
 * clear: the folded GEMM followed by hg_test_attention's kernel - the two must agree bit for bit.  fused bit 1 set: a is
 * [n_seq*L, D + 64] and w [3D, D + 64], the summed length of a block whose adapter is folded into in_proj (variant C). */
int hg_test_qkv_attn(hg_ctx*, const float* a, const float* w, const float* bias, const float* cs, const float* mr, int n_seq,
                     int L, int heads, int fused, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HOIGEN_AMD_H */
