// C ABI of hoigen_amd (include/hoigen_amd.h): context, weight conversion into the device layout,
// grow-only workspace and the launch sequences of the three hot paths (image tower, text tower,
// CoOp-VAE).  Host code only; kernels live in hg_gemm.hip / hg_attn.hip / hg_elem.hip / hg_adapter.hip.
//
// Device data layout:
//   residual stream x   fp32 [M, D]      M = n_seq * L rows (token-major, sequence-contiguous)
//   h / att / fc        fp16 [M, D|4D]   MFMA A operands (K contiguous)
//   qkv                 fp16 [M, 3D]     q|k|v column blocks, head h = columns 64h..64h+63
//   linear weights      fp16 [N, K]      exactly nn.Linear's [out, in] -> both GEMM operands K-contiguous
//   proj/text_projection fp16 [E, D]     transposed once at load ([D,E] in the state dict)
//   biases, LN affine, embeddings, positional: fp32
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include <algorithm>
#include <mutex>

#include "../../include/hoigen_amd.h"
#include "hg_kernels.h"

using namespace hg;

namespace {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct BlockW {
    half_t *w_qkv, *w_out, *w_fc, *w_proj;
    float *b_qkv, *b_out, *b_fc, *b_proj, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    // LayerNorm folded into the consuming GEMM: W' = fp16(W * gamma), cs[n] = sum_k W'[n][k],
    // b' = b + W beta;  LN(x) W^T + b = rstd * (x16 W'^T - mean * cs) + b'
    half_t *wf_qkv, *wf_fc;
    float *cs_qkv, *bf_qkv, *cs_fc, *bf_fc;
    // column sums sum_k gamma[k] W[n][k] for the form that keeps the LayerNorm weight in the activation copy and W unrounded (text tower)
    float *csg_qkv, *csg_fc;
    // wf_qkv / bf_qkv / cs_qkv once more in the order the fused in_proj + attention kernel streams them (hg_qkv_attn.hip:
    // MFMA fragments per head pair); null when the width does not qualify
    half_t* wp_qkv;
    float* bcs_qkv;
    // c_fc / c_proj as the fragment stream of the one-kernel MLP block (hg_vae_fused.hip, mode 3): width 512 only (the text tower)
    half_t* wp_mlp;
};

struct AdapterW {
    bool present = false;
    int d = 0;
    half_t* down_w = nullptr;  // [128 (padded), D]
    float* down_b = nullptr;   // [128]
    float* down_cs = nullptr;  // [128] row sums of the fp16 weight: down_proj on the CENTRED fp16 copy adds mu * cs back
    half_t* up_w = nullptr;    // [D, 64]
    float* up_b = nullptr;
    float* scale = nullptr;
    // The adapter folded into the block's GEMMs altogether (hg_elem.hip adapter_q_kernel): a = Q e with e = the last decoder
    // layer's normalised output; [0] = prior path (last layer of the mhsa_layers chain), [1] = self path (mhsa)
    struct Fold {
        half_t* down2 = nullptr;    // [128, D]: down_proj rows | Q^T (the cross term of the statistics rides in the padded half)
        half_t* wk_out = nullptr;   // [D, D + 64] = [W_out | Q]
        half_t* wq_cat = nullptr;   // [3D, D + 64] = [W'_qkv | W'_qkv Q]
        half_t* wp_qcat = nullptr;  // wq_cat in the fused in_proj + attention kernel's fragment order (launch_pack_qkv, K = D + 64)
        half_t* g16 = nullptr;      // [64, 64] Q^T Q
        float* qm = nullptr;        // [64] column sums of Q
    } fold[2];
    float* dl[2][12] = {};     // see AdapterDev
    half_t* w16[2][6] = {};    // see AdapterDev
    struct Extra { float* dl[12] = {}; half_t* w16[6] = {}; };
    std::vector<Extra> extra;  // mhsa_layers.1 .. N-1 (adapter_num_layers > 1), prior path only
};

struct Vit {
    bool loaded = false;
    int D = 0, layers = 0, heads = 0, patch = 0, res = 0, grid = 0, L = 0, E = 0, Kp = 0;
    half_t* w_patch = nullptr;
    float *cls = nullptr, *pos = nullptr, *lnpre_w = nullptr, *lnpre_b = nullptr, *lnpost_w = nullptr,
          *lnpost_b = nullptr;
    half_t* w_projT = nullptr;
    std::vector<BlockW> blocks;
    std::vector<AdapterW> adapters;
    std::vector<void*> owned, owned_adapters;
};

struct Text {
    bool loaded = false;
    int D = 0, layers = 0, heads = 0, ctx = 0, vocab = 0, E = 0;
    float *tok = nullptr, *pos = nullptr, *lnf_w = nullptr, *lnf_b = nullptr;
    half_t* w_projT = nullptr;
    std::vector<BlockW> blocks;
    std::vector<void*> owned;
};

struct Vae {
    bool enc = false, gen = false;
    int dim = 0, eh = 0, gh = 0;
    half_t *e_w0 = nullptr, *e_wml = nullptr, *g_w0 = nullptr, *g_w2 = nullptr;
    float *e_b0 = nullptr, *e_bml = nullptr, *g_b0 = nullptr, *g_b2 = nullptr;
    half_t* wp = nullptr;      // the same weights as the fragment stream of the one-kernel path (hg_vae_fused.hip): [E0 | E1 | G]
    // the same stacked mean | log_var operand with its rows interleaved in blocks of 128 (EPI_VAE_REPARAM_F32)
    std::vector<void*> owned;
};

struct Cache {
    bool loaded = false, has_labels = false;
    int S = 0, K = 0, C = 0, Sp = 0, Cp = 0;
    half_t *w16 = nullptr, *lt16 = nullptr;   // [Sp,K] ; labels^T [Cp,Sp]
    float *b = nullptr, *bias_c = nullptr, *scale = nullptr;   // [Sp] ; [Cp] bias @ labels ; [Cp] 1 / (lens * post_div)
    std::vector<void*> owned;
};

struct Mlp {
    bool loaded = false;
    int in = 0, hid = 0, out = 0;
    half_t *w0 = nullptr, *w2 = nullptr, *w4 = nullptr;
    float *b0 = nullptr, *b2 = nullptr, *b4 = nullptr;
    std::vector<void*> owned;
};

}  // namespace

struct hg_ctx {
    int device = 0;
    std::string err;
    Vit vit;
    Text text;
    Vae vae[HG_MAX_SLOTS];
    Mlp mlp[HG_MAX_SLOTS];
    Cache cache[HG_MAX_CACHE_SLOTS];
    // workspace (grow-only)
    Buf x, h, qkv, att, fc, head16, tok32, small, i32, ad32, ad16, adkv, mr, mu, muc, stats, pre, pretab, cx, ca, ch, cf, cq;
    Buf hg;              // folded path with the LayerNorm weight in the activation copy (text tower): that copy, beside the stream's hi half in h
    Buf att2;            // variant C with the stream as centre + hi + lo: the out-proj operand [att | e] beside the in_proj one [x16 | e]
    Buf zpark;           // hg_vae_fused.hip: the encoder's first z half as fp16 fragments, per wave
    Buf xlo;             // low half of the residual stream while it is held as centre + hi + lo (GemmArgs::hl)
    Buf pair_ready;      // hg_mlp_pair.hip: ready counters [blocks][256-row panels], zeroed at the start of every tower pass
    int max_chunk_img = 256;
    int text_rows_budget = 65536;      // rows (prompts x executed tokens) per pass of the text tower (text_chunk_prompts)
    int max_chunk_rows = 32768;
    // behaviour options: hg_set_option; the environment (HG_LAST_BLOCK_ROW0, HG_LN_FUSE, HG_ADAPTER_FUSE, HG_ADAPTER_FOLD,
    // HG_CHUNK_ROWS) only gives their values at hg_create - nothing on the call path reads the environment
    int opt_row0 = 1;            // last block of a tower without token outputs on the one row that leaves it
    int opt_ln_fuse = 1;         // LayerNorm folded into the GEMMs where the shapes allow
    int opt_adapter_fuse = 1;    // ... also behind the instance adapters (variant C)
    int opt_adapter_fold = 1;    // adapter folded into the block's own QKV / out-proj GEMMs (0: separate up_proj GEMM)
    int opt_stream_hilo = 1;     // residual stream as centre + hi + lo (fp16 copy + bf8 remainder) between the folded blocks (0: fp32)
    int opt_qkv_attn = 1;        // in_proj + attention as one kernel, q / k / v kept in LDS (vision tower, folded blocks; 0: two kernels)
    int opt_qkv_attn_min_seq = 32;   // ... from this many sequences per call on, and where its last round of items is filled well
                                     // enough (qkv_attn_pays; qkv_attn = 2: wherever the shapes allow)
    int opt_qkv_attn_gsz = 0;    // head pairs per XCD group of that kernel (0 = all)
    int opt_text_ln_fold = 1;    // text tower: 1 (default) LayerNorm folded into its GEMMs with the LayerNorm weight in the ACTIVATION copy (GemmArgs::gamma:
                                 // the GEMMs keep the layer's own fp16 weights - closer to the reference than the separate kernels, 4 % faster);
                                 // 2 the weight folded into fp16(W * gamma) as in the vision tower (10 % faster, 7.6e-4 instead of 6.2e-4); 0 separate kernels
    int opt_qkv_attn_c = 1;      // ... also in the blocks that carry a folded adapter (variant C on the hi / lo stream: K = D + 64)
    int opt_mlp_fused = 0;       // blocks of width 512 (text tower): 1 = c_fc -> QuickGELU -> c_proj -> residual as ONE kernel for the rows that
                                 // fill whole rounds of 128-row items (hg_vae_fused.hip, mode 3), 2 = every row, 0 (default) = the two GEMMs:
                                 // measured a tie (65 534-row pass: 331-335 us against 161 + 190; the generation pipeline -1.4 %) and a loss
                                 // where a call leaves a remainder (600 prompts x 77 tokens: 5.47 -> 5.66 ms)
    int opt_vae_fused = 1;       // CoOp-VAE Encoder -> reparameterise -> Generator as ONE kernel (hg_vae_fused.hip) for the rows that fill
                                 // whole rounds of 128-row items over the CUs (the rest: the GEMM path); 2: every row; 0: GEMM path only
    int opt_mlp_pair = 1;        // c_fc -> QuickGELU -> c_proj of a LayerNorm-folded block as ONE persistent launch with per-row-panel ready
                                 // counters between its tiles (hg_mlp_pair.hip; both towers, variant A); bit-identical to the two launches;
                                 // 2: finalize_stats of the next LayerNorm in the launch's tail as well
    int opt_mlp_pair_chunk = 32; // ... 256-row panels of an XCD per chunk
    int opt_mlp_pair_fc_slots = 32;  // ... workgroups per XCD that run c_fc tiles (the rest start with c_proj)
    int opt_mlp_pair_fault = 0;      // fault injection for the tests: that launch goes out one workgroup short, so that a hand-off wait meets its bound
    int n_cu = 256;
    // sticky device->host flag (host-mapped): a hand-off wait inside the MLP pair kernel gave up (a workgroup of its grid never became
    // resident); the call in flight returned garbage, the next tower call reports HG_ERR_HIP
    int32_t* pair_err = nullptr;
    // sticky device->host flag (host-mapped): inside a tower with folded LayerNorms a row reached further from its centre than the centred
    // fp16 copy / the hi half of the stream can hold (|x - row centre| > 65504, bounded through the row statistics: finalize_stats).
    // Reported as HG_ERR_INVALID by the next tower call.
    int32_t* range_flag = nullptr;
    // sticky device->host flag (host-mapped): set by clamp_eot when a caller-supplied text truncation was shorter than
    // max(EOT)+1 (a stale host memo); reported as HG_ERR_INVALID by the next text call
    int32_t* eot_flag = nullptr;
    int32_t* eot_flag_dev = nullptr;   // the same condition for the call in flight, in device memory (zeroed per call): poison_if_flag polls it
    // live per-kernel timing for bench.py (hg_profile_begin/end): hipEvent pairs around the launches of one kernel
    // kind (or of every GEMM and attention launch), on the stream the kernel is launched on
    int prof_kind = HG_PROF_OFF;
    std::vector<hipEvent_t> prof_ev;
    std::vector<hg_prof_rec> prof_rec;
    size_t prof_n = 0;
};

namespace {

int fail(hg_ctx* c, int code, const char* fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (c) c->err = tmp;
    return code;
}

#define HG_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(c, HG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                     \
    } while (0)

// Entry points run on the context's device and give the caller's current device back on return (torch tracks the
// current device per thread; a library that silently changes it redirects the caller's next allocation).
struct DevGuard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DevGuard(const hg_ctx* c) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) err = hipSetDevice(c->device);
        else prev = -1;
    }
    ~DevGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
#define HG_ON_DEVICE(c)                                                                                 \
    DevGuard dev_guard_(c);                                                                             \
    if (dev_guard_.err != hipSuccess)                                                                   \
        return fail(c, HG_ERR_HIP, "hipSetDevice(%d) failed: %s", (c)->device, hipGetErrorString(dev_guard_.err))
// first failing status wins (OR-ing negative codes can turn OOM into another code)
inline void keep_first(int& rc, int r) {
    if (!rc) rc = r;
}

// The MLP pair launch (hg_mlp_pair.hip) has one workgroup per CU and its workgroups wait for each other: two such launches from different
// streams of this process, each holding part of the CUs, would wait for each other until their bound (both towers of a model run it, and a
// caller may well encode text on one stream and crops on another).  As long as every pair launch of a device comes from ONE stream
// nothing is done.  The first time a second stream shows up the device is synchronised once, and from then on every pair launch waits for
// the previous one's event and records its own: the launches are serial across streams (each fills the chip anyway).  Other processes
// on the same GPU are the deployment's business (option mlp_pair = 0 there).
struct PairGate {
    std::mutex mu;
    bool have_first = false, multi = false, evt_set = false;
    hipStream_t first = nullptr;
    hipEvent_t evt = nullptr;
};
static PairGate g_pair_gate[16];
struct PairGateScope {      // around ONE pair launch on stream s of device dev (the current device)
    PairGate* g;
    hipStream_t s;
    PairGateScope(int dev, hipStream_t s_) : g(&g_pair_gate[dev & 15]), s(s_) {
        g->mu.lock();
#ifdef HG_NO_PAIR_GATE      // (experiment build: what two streams do to each other without the gate)
        return;
#endif
        if (!g->have_first) { g->first = s; g->have_first = true; }
        else if (!g->multi && s != g->first) {
            (void)hipDeviceSynchronize();      // (once: whatever the first stream has in flight carries no event)
            if (hipEventCreateWithFlags(&g->evt, hipEventDisableTiming) != hipSuccess) g->evt = nullptr;
            g->multi = true;
        }
        if (g->multi && g->evt && g->evt_set) (void)hipStreamWaitEvent(s, g->evt, 0);
    }
    ~PairGateScope() {
        if (g->multi && g->evt && hipEventRecord(g->evt, s) == hipSuccess) g->evt_set = true;
        g->mu.unlock();
    }
};

int ensure(hg_ctx* c, Buf& b, size_t bytes) {
    if (b.bytes >= bytes) return HG_OK;
    if (b.p) HG_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    bytes = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) return fail(c, HG_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    // growth only (never in steady state): zero the padding rows and order the memset against every stream, also
    // non-blocking ones that do not synchronise with the null stream
    HG_HIP(hipMemset(b.p, 0, bytes));
    HG_HIP(hipDeviceSynchronize());
    b.bytes = bytes;
    return HG_OK;
}

void free_all(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

// ---- weight conversion helpers (synchronous; load time only) ------------------------------------
int dev_alloc(hg_ctx* c, std::vector<void*>& owned, size_t bytes, void** out) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(c, HG_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    owned.push_back(p);
    *out = p;
    return HG_OK;
}

int as_f16(hg_ctx* c, std::vector<void*>& owned, const hg_tensor& t, size_t n, half_t** out, const char* name) {
    if (!t.ptr) return fail(c, HG_ERR_INVALID, "missing tensor %s", name);
    void* p;
    int rc = dev_alloc(c, owned, n * 2, &p);
    if (rc) return rc;
    if (t.dtype == HG_F16) HG_HIP(hipMemcpy(p, t.ptr, n * 2, hipMemcpyDeviceToDevice));
    else if (t.dtype == HG_F32) HG_HIP(launch_f32_to_f16((const float*)t.ptr, (half_t*)p, n, 0));
    else return fail(c, HG_ERR_INVALID, "bad dtype for %s", name);
    *out = (half_t*)p;
    return HG_OK;
}

int as_f32(hg_ctx* c, std::vector<void*>& owned, const hg_tensor& t, size_t n, float** out, const char* name) {
    if (!t.ptr) return fail(c, HG_ERR_INVALID, "missing tensor %s", name);
    void* p;
    int rc = dev_alloc(c, owned, n * 4, &p);
    if (rc) return rc;
    if (t.dtype == HG_F32) HG_HIP(hipMemcpy(p, t.ptr, n * 4, hipMemcpyDeviceToDevice));
    else if (t.dtype == HG_F16) HG_HIP(launch_f16_to_f32((const half_t*)t.ptr, (float*)p, n, 0));
    else return fail(c, HG_ERR_INVALID, "bad dtype for %s", name);
    *out = (float*)p;
    return HG_OK;
}

// [rows, cols] -> fp16 [cols, rows]
int as_f16_T(hg_ctx* c, std::vector<void*>& owned, const hg_tensor& t, int rows, int cols, half_t** out,
             const char* name) {
    if (!t.ptr) return fail(c, HG_ERR_INVALID, "missing tensor %s", name);
    void* p;
    int rc = dev_alloc(c, owned, (size_t)rows * cols * 2, &p);
    if (rc) return rc;
    HG_HIP(launch_transpose_to_f16(t.ptr, t.dtype, (half_t*)p, rows, cols, 0));
    *out = (half_t*)p;
    return HG_OK;
}

// fp32 [rows, cols] -> fp32 [cols, rows] via host (tiny adapter matrices)
int as_f32_T(hg_ctx* c, std::vector<void*>& owned, const hg_tensor& t, int rows, int cols, float** out,
             const char* name) {
    float* tmp;
    std::vector<void*> scratch;
    int rc = as_f32(c, scratch, t, (size_t)rows * cols, &tmp, name);
    if (rc) { free_all(scratch); return rc; }
    std::vector<float> h((size_t)rows * cols), ht((size_t)rows * cols);
    hipError_t e = hipMemcpy(h.data(), tmp, h.size() * 4, hipMemcpyDeviceToHost);
    free_all(scratch);
    if (e != hipSuccess) return fail(c, HG_ERR_HIP, "hipMemcpy D2H failed for %s", name);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cols; ++k) ht[(size_t)k * rows + r] = h[(size_t)r * cols + k];
    void* p;
    rc = dev_alloc(c, owned, ht.size() * 4, &p);
    if (rc) return rc;
    HG_HIP(hipMemcpy(p, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return HG_OK;
}

// The MLP of a width-512 block as one kernel (option mlp_fused, hg_vae_fused.hip mode 3) reads its two weights as a packed fragment
// stream: 4.3 MB per block, 52 MB for the text tower - packed when the option is on at load time or is switched on later, not for
// every tower (the option defaults to 0: measured a tie).
int pack_mlp_blocks(hg_ctx* c, std::vector<void*>& owned, std::vector<BlockW>& blocks, int D) {
    if (!vae_fused_ok(D, 0, 4 * D)) return HG_OK;
    for (BlockW& b : blocks) {
        if (b.wp_mlp) continue;
        int rc = 0;
        keep_first(rc, dev_alloc(c, owned, vae_fused_pass_bytes(4 * D), (void**)&b.wp_mlp));
        if (rc) return rc < 0 ? rc : HG_ERR_OOM;
        HG_HIP(launch_pack_vae(nullptr, nullptr, 0, b.w_fc, b.w_proj, 4 * D, b.wp_mlp, 0));
    }
    return HG_OK;
}

int load_blocks(hg_ctx* c, std::vector<void*>& owned, const hg_block_weights* src, int layers, int D,
                std::vector<BlockW>& dst, bool fold_ln) {
    if (!src) return fail(c, HG_ERR_INVALID, "blocks == NULL");
    dst.assign(layers, BlockW{});
    for (int i = 0; i < layers; ++i) {
        const hg_block_weights& s = src[i];
        BlockW& b = dst[i];
        int rc = 0;
        keep_first(rc, as_f16(c, owned, s.in_proj_weight, (size_t)3 * D * D, &b.w_qkv, "attn.in_proj_weight"));
        keep_first(rc, as_f32(c, owned, s.in_proj_bias, (size_t)3 * D, &b.b_qkv, "attn.in_proj_bias"));
        keep_first(rc, as_f16(c, owned, s.out_proj_weight, (size_t)D * D, &b.w_out, "attn.out_proj.weight"));
        keep_first(rc, as_f32(c, owned, s.out_proj_bias, D, &b.b_out, "attn.out_proj.bias"));
        keep_first(rc, as_f32(c, owned, s.ln_1_weight, D, &b.ln1_w, "ln_1.weight"));
        keep_first(rc, as_f32(c, owned, s.ln_1_bias, D, &b.ln1_b, "ln_1.bias"));
        keep_first(rc, as_f16(c, owned, s.c_fc_weight, (size_t)4 * D * D, &b.w_fc, "mlp.c_fc.weight"));
        keep_first(rc, as_f32(c, owned, s.c_fc_bias, (size_t)4 * D, &b.b_fc, "mlp.c_fc.bias"));
        keep_first(rc, as_f16(c, owned, s.c_proj_weight, (size_t)4 * D * D, &b.w_proj, "mlp.c_proj.weight"));
        keep_first(rc, as_f32(c, owned, s.c_proj_bias, D, &b.b_proj, "mlp.c_proj.bias"));
        keep_first(rc, as_f32(c, owned, s.ln_2_weight, D, &b.ln2_w, "ln_2.weight"));
        keep_first(rc, as_f32(c, owned, s.ln_2_bias, D, &b.ln2_b, "ln_2.bias"));
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        if (!fold_ln) continue;
        keep_first(rc, dev_alloc(c, owned, (size_t)3 * D * D * 2, (void**)&b.wf_qkv));
        keep_first(rc, dev_alloc(c, owned, (size_t)3 * D * 4, (void**)&b.cs_qkv));
        keep_first(rc, dev_alloc(c, owned, (size_t)3 * D * 4, (void**)&b.csg_qkv));
        keep_first(rc, dev_alloc(c, owned, (size_t)4 * D * 4, (void**)&b.csg_fc));
        keep_first(rc, dev_alloc(c, owned, (size_t)3 * D * 4, (void**)&b.bf_qkv));
        keep_first(rc, dev_alloc(c, owned, (size_t)4 * D * D * 2, (void**)&b.wf_fc));
        keep_first(rc, dev_alloc(c, owned, (size_t)4 * D * 4, (void**)&b.cs_fc));
        keep_first(rc, dev_alloc(c, owned, (size_t)4 * D * 4, (void**)&b.bf_fc));
        if (rc) return rc < 0 ? rc : HG_ERR_OOM;
        HG_HIP(launch_fold_ln(b.w_qkv, b.ln1_w, b.ln1_b, b.b_qkv, b.wf_qkv, b.cs_qkv, b.bf_qkv, 3 * D, D, 0, b.csg_qkv));
        HG_HIP(launch_fold_ln(b.w_fc, b.ln2_w, b.ln2_b, b.b_fc, b.wf_fc, b.cs_fc, b.bf_fc, 4 * D, D, 0, b.csg_fc));
        if (qkv_attn_ok(1, 197, D, D / 64, D)) {      // (heads = width / 64 in every CLIP tower; L is checked per call)
            keep_first(rc, dev_alloc(c, owned, (size_t)3 * D * D * 2, (void**)&b.wp_qkv));
            keep_first(rc, dev_alloc(c, owned, (size_t)(D / 128) * 768 * 4, (void**)&b.bcs_qkv));
            if (rc) return rc < 0 ? rc : HG_ERR_OOM;
            HG_HIP(launch_pack_qkv(b.wf_qkv, b.bf_qkv, b.cs_qkv, b.wp_qkv, b.bcs_qkv, D, D / 64, 0));
        }

    }
    if (c->opt_mlp_fused) {
        int rc = pack_mlp_blocks(c, owned, dst, D);
        if (rc) return rc;
    }
    return HG_OK;
}

int load_decoder_layer(hg_ctx* c, std::vector<void*>& owned, const hg_decoder_layer_weights& s, int d,
                       float* dl[12], half_t* w16[6]) {
    {      // fp16 [out][in] operands of the MFMA decoder: the state dict's own layout
        half_t* inw16 = nullptr;
        int r16 = as_f16(c, owned, s.attn_in_proj_weight, (size_t)3 * d * d, &inw16, "adapter in_proj_weight");
        if (!r16) { w16[0] = inw16; w16[1] = inw16 + (size_t)d * d; w16[2] = inw16 + (size_t)2 * d * d; }
        if (!r16) r16 = as_f16(c, owned, s.attn_out_proj_weight, (size_t)d * d, &w16[3], "adapter out_proj.weight");
        if (!r16) r16 = as_f16(c, owned, s.linear1_weight, (size_t)2 * d * d, &w16[4], "adapter linear1.weight");
        if (!r16) r16 = as_f16(c, owned, s.linear2_weight, (size_t)2 * d * d, &w16[5], "adapter linear2.weight");
        if (r16) return r16;
    }
    // 0 WqT [d,d] (in->out), 1 bq, 2 WkT, 3 bk, 4 WvT, 5 bv : split of in_proj;  6 WoT, 7 bo ... see below
    std::vector<void*> scratch;
    float* inw;
    float* inb;
    int rc = as_f32(c, scratch, s.attn_in_proj_weight, (size_t)3 * d * d, &inw, "adapter in_proj_weight");
    if (!rc) rc = as_f32(c, scratch, s.attn_in_proj_bias, (size_t)3 * d, &inb, "adapter in_proj_bias");
    if (rc) { free_all(scratch); return rc; }
    for (int part = 0; part < 3 && !rc; ++part) {
        hg_tensor wt{inw + (size_t)part * d * d, HG_F32};
        hg_tensor bt{inb + (size_t)part * d, HG_F32};
        rc = as_f32_T(c, owned, wt, d, d, &dl[part], "adapter q/k/v weight");
        if (!rc) rc = as_f32(c, owned, bt, d, &dl[3 + part], "adapter q/k/v bias");
    }
    free_all(scratch);
    if (rc) return rc;
    keep_first(rc, as_f32_T(c, owned, s.attn_out_proj_weight, d, d, &dl[6], "adapter out_proj.weight"));
    keep_first(rc, as_f32(c, owned, s.attn_out_proj_bias, d, &dl[7], "adapter out_proj.bias"));
    // norm2 | norm3 packed: [w2, b2, w3, b3] (4*d)
    {
        float* p;
        int r2 = dev_alloc(c, owned, (size_t)4 * d * 4, (void**)&p);
        if (r2) return r2;
        const hg_tensor* ts[4] = {&s.norm2_weight, &s.norm2_bias, &s.norm3_weight, &s.norm3_bias};
        for (int k = 0; k < 4; ++k) {
            float* t;
            std::vector<void*> sc;
            int r3 = as_f32(c, sc, *ts[k], d, &t, "adapter norm");
            if (r3) { free_all(sc); return r3; }
            hipError_t e = hipMemcpy(p + (size_t)k * d, t, (size_t)d * 4, hipMemcpyDeviceToDevice);
            free_all(sc);
            if (e != hipSuccess) return fail(c, HG_ERR_HIP, "memcpy norm failed");
        }
        dl[8] = p;
    }
    keep_first(rc, as_f32_T(c, owned, s.linear1_weight, 2 * d, d, &dl[9], "adapter linear1.weight"));   // [d, 2d]
    keep_first(rc, as_f32(c, owned, s.linear1_bias, (size_t)2 * d, &dl[10], "adapter linear1.bias"));
    // linear2: weight^T [2d, d] followed by bias [d]
    {
        float* w2t;
        std::vector<void*> sc;
        int r2 = as_f32_T(c, sc, s.linear2_weight, d, 2 * d, &w2t, "adapter linear2.weight");
        float* b2 = nullptr;
        if (!r2) r2 = as_f32(c, sc, s.linear2_bias, d, &b2, "adapter linear2.bias");
        float* p = nullptr;
        if (!r2) r2 = dev_alloc(c, owned, ((size_t)2 * d * d + d) * 4, (void**)&p);
        if (!r2) {
            hipError_t e = hipMemcpy(p, w2t, (size_t)2 * d * d * 4, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(p + (size_t)2 * d * d, b2, (size_t)d * 4, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) r2 = fail(c, HG_ERR_HIP, "memcpy linear2 failed");
        }
        free_all(sc);
        if (r2) return r2;
        dl[11] = p;
    }
    return rc ? (rc < 0 ? rc : HG_ERR_INVALID) : HG_OK;
}

int upload_f32(hg_ctx* c, std::vector<void*>& owned, const std::vector<float>& v, float** out);

int load_adapters(hg_ctx* c, const hg_adapter_weights* src, int layers) {
    Vit& v = c->vit;
    free_all(v.owned_adapters);
    v.adapters.assign(v.layers, AdapterW{});
    if (!src) return HG_OK;
    if (layers != v.layers) return fail(c, HG_ERR_INVALID, "adapter layer count %d != %d", layers, v.layers);
    const int D = v.D;
    for (int i = 0; i < layers; ++i) {
        const hg_adapter_weights& s = src[i];
        if (!s.present) continue;
        if (s.bottleneck != 64) return fail(c, HG_ERR_INVALID, "adapter bottleneck must be 64 (got %d)", s.bottleneck);
        AdapterW& a = v.adapters[i];
        const int d = 64;
        a.d = d;
        std::vector<void*>& own = v.owned_adapters;
        // down_proj padded to 128 output rows (the GEMM tile is 128 wide); rows 64.. are zero
        void* p;
        int rc = dev_alloc(c, own, (size_t)128 * D * 2, &p);
        if (rc) return rc;
        HG_HIP(hipMemset(p, 0, (size_t)128 * D * 2));
        a.down_w = (half_t*)p;
        if (!s.down_proj_weight.ptr) return fail(c, HG_ERR_INVALID, "missing adapter down_proj.weight");
        if (s.down_proj_weight.dtype == HG_F16)
            HG_HIP(hipMemcpy(p, s.down_proj_weight.ptr, (size_t)d * D * 2, hipMemcpyDeviceToDevice));
        else HG_HIP(launch_f32_to_f16((const float*)s.down_proj_weight.ptr, a.down_w, (size_t)d * D, 0));
        rc = dev_alloc(c, own, 128 * 4, &p);
        if (rc) return rc;
        HG_HIP(hipMemset(p, 0, 128 * 4));
        a.down_b = (float*)p;
        {
            float* t;
            std::vector<void*> sc;
            rc = as_f32(c, sc, s.down_proj_bias, d, &t, "adapter down_proj.bias");
            if (!rc && hipMemcpy(p, t, d * 4, hipMemcpyDeviceToDevice) != hipSuccess) rc = HG_ERR_HIP;
            free_all(sc);
            if (rc) return rc;
        }
        {      // cs[n] = sum_k float(W16[n][k]) through the LayerNorm-folding helper with gamma = 1, beta = 0
            std::vector<float> ones(D, 1.0f), zeros(D, 0.0f);
            std::vector<void*> sc;
            float *g1 = nullptr, *b0 = nullptr, *bf = nullptr;
            half_t* wf = nullptr;
            int r2 = upload_f32(c, sc, ones, &g1);
            if (!r2) r2 = upload_f32(c, sc, zeros, &b0);
            if (!r2) r2 = dev_alloc(c, sc, (size_t)128 * D * 2, (void**)&wf);
            if (!r2) r2 = dev_alloc(c, sc, 128 * 4, (void**)&bf);
            if (!r2) r2 = dev_alloc(c, own, 128 * 4, (void**)&a.down_cs);
            if (!r2) {
                hipError_t e = launch_fold_ln(a.down_w, g1, b0, a.down_b, wf, a.down_cs, bf, 128, D, 0);
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (e != hipSuccess) r2 = fail(c, HG_ERR_HIP, "down_proj row sums failed: %s", hipGetErrorString(e));
            }
            free_all(sc);
            if (r2) return r2;
        }
        keep_first(rc, as_f16(c, own, s.up_proj_weight, (size_t)D * d, &a.up_w, "adapter up_proj.weight"));
        keep_first(rc, as_f32(c, own, s.up_proj_bias, D, &a.up_b, "adapter up_proj.bias"));
        keep_first(rc, as_f32(c, own, s.scale, D, &a.scale, "adapter scale"));
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        rc = load_decoder_layer(c, own, s.prior_layer, d, a.dl[0], a.w16[0]);
        if (rc) return rc;
        rc = load_decoder_layer(c, own, s.self_layer, d, a.dl[1], a.w16[1]);
        if (rc) return rc;
        if (s.n_extra_prior_layers < 0 || (s.n_extra_prior_layers > 0 && !s.extra_prior_layers))
            return fail(c, HG_ERR_INVALID, "adapter %d: bad extra_prior_layers", i);
        a.extra.resize(s.n_extra_prior_layers);
        for (int z = 0; z < s.n_extra_prior_layers; ++z) {
            rc = load_decoder_layer(c, own, s.extra_prior_layers[z], d, a.extra[z].dl, a.extra[z].w16);
            if (rc) return rc;
        }
        if ((int)v.blocks.size() > i && v.blocks[i].w_out && v.blocks[i].wf_qkv) {
            std::vector<void*> sc;
            float* q32 = nullptr;
            keep_first(rc, dev_alloc(c, sc, (size_t)D * d * 4, (void**)&q32));
            for (int k = 0; k < 2 && !rc; ++k) {
                AdapterW::Fold& f = a.fold[k];
                keep_first(rc, dev_alloc(c, own, (size_t)128 * D * 2, (void**)&f.down2));
                keep_first(rc, dev_alloc(c, own, (size_t)D * (D + d) * 2, (void**)&f.wk_out));
                keep_first(rc, dev_alloc(c, own, (size_t)3 * D * (D + d) * 2, (void**)&f.wq_cat));
                keep_first(rc, dev_alloc(c, own, (size_t)d * d * 2, (void**)&f.g16));
                keep_first(rc, dev_alloc(c, own, (size_t)d * 4, (void**)&f.qm));
                if (rc) break;
                const float* norms = k == 0 ? (a.extra.empty() ? a.dl[0][8] : a.extra.back().dl[8]) : a.dl[1][8];
                hipError_t e = launch_adapter_fold(a.up_w, a.up_b, a.scale, norms, a.down_w, v.blocks[i].w_out, v.blocks[i].wf_qkv, D,
                                                   q32, f.down2, f.wk_out, f.wq_cat, f.qm, f.g16, 0);
                if (e == hipSuccess && d == 64 && qkv_attn_ok(1, 197, D, D / 64, D + 64, D + 64)) {
                    keep_first(rc, dev_alloc(c, own, (size_t)3 * D * (D + d) * 2, (void**)&f.wp_qcat));
                    if (rc) break;
                    e = launch_pack_qkv(f.wq_cat, nullptr, nullptr, f.wp_qcat, nullptr, D, D / 64, 0, D + 64);
                }
                if (e == hipSuccess) e = hipDeviceSynchronize();
                if (e != hipSuccess) rc = fail(c, HG_ERR_HIP, "adapter fold failed: %s", hipGetErrorString(e));
            }
            free_all(sc);
            if (rc) return rc;
        }
        a.present = true;
    }
    HG_HIP(hipDeviceSynchronize());
    return HG_OK;
}

inline size_t rup(size_t v, size_t m) { return (v + m - 1) / m * m; }

// Rows of the next chunk of a row-parallel call with `left` rows to go: max_chunk_rows, except that the LAST chunk absorbs a
// tail of up to an eighth of it (100 000 rows = 32 768 + 32 768 + 34 464 instead of + 32 768 + 1 696: the four dependent GEMMs of a
// 1 696-row chunk fill 14-56 tiles each and cost 0.12 ms for 1.7 % of the rows; workspace +5 %)
inline int chunk_rows(const hg_ctx* c, int left) {
    const int m = c->max_chunk_rows;
    return left <= m + m / 8 ? left : m;
}

// Rows (from row 0) that go to the one-kernel path: its work items are 128 rows and take 0.3-0.4 ms each, so it only pays for whole
// rounds of items over the CUs (100 000 rows = 782 items = 3 rounds of 256 + 14: the 14 would cost a fourth round); the rest - and calls
// too small to fill most of one round - take the GEMM path, whose 256 x 256 tiles quantise a hundred times finer.
inline int fused_item_rows(const hg_ctx* c, int opt, int R) {
    if (opt == 0 || R <= 0) return 0;
    if (opt == 2) return R;
    const int per = vae_fused_rows_per_item();
    const long items = ((long)R + per - 1) / per, ncu = c->n_cu;
    const long full = items / ncu * ncu, rem = items - full;
    const long take = full + (rem * 100 >= ncu * 70 ? rem : 0);
    const long rows = take * per;
    return (int)(rows < R ? rows : R);
}
// hipEvent pair around one launch when its kind is being profiled
struct ProfScope {
    hg_ctx* c;
    hipStream_t s;
    bool on;
    ProfScope(hg_ctx* c_, hipStream_t s_, int kind, int M, int N, int K) : c(c_), s(s_), on(false) {
        if (c->prof_kind == HG_PROF_OFF || (c->prof_kind != HG_PROF_ALL && c->prof_kind != kind)) return;
        if (2 * c->prof_n + 1 >= c->prof_ev.size()) return;
        if (hipEventRecord(c->prof_ev[2 * c->prof_n], s) != hipSuccess) return;
        c->prof_rec[c->prof_n] = hg_prof_rec{kind, M, N, K, 0.f};
        on = true;
    }
    void finish() {
        if (on && hipEventRecord(c->prof_ev[2 * c->prof_n + 1], s) == hipSuccess) c->prof_n++;
        on = false;
    }
    ~ProfScope() { finish(); }
};

hipError_t gemm(hg_ctx* c, int epi, const GemmArgs& g, hipStream_t s) {
    ProfScope ps(c, s, epi, g.M, g.N, g.K);
    return launch_gemm(epi, g, s);
}

hipError_t attention(hg_ctx* c, const half_t* qkv, half_t* out, int n_seq, int L, int heads, bool causal, hipStream_t s,
                     int ldo = 0) {
    ProfScope ps(c, s, HG_PROF_ATTENTION, n_seq, L, heads);
    return launch_attention(qkv, out, n_seq, L, heads, causal, s, ldo);
}

// ---- one transformer tower over the residual stream in c->x ------------------------------------------
struct AdapterCall {
    const float* priors = nullptr;   // [n_seq, N, 64] or null
    const uint8_t* mask = nullptr;
    int N = 0;
    bool enabled = false;
};

// `fused`: LayerNorm folding is on - the stream's centred fp16 copy (c->h), its centre (c->muc) and the folding
// statistics are current; the adapter consumes the copy and its up_proj re-emits all three for the updated stream
// `kcat` = 2 (0 = off): the adapter folded into the block's GEMMs altogether - nothing but down_proj and the decoder runs here; the
// centred fp16 copy of the stream is expected in columns 0..D-1 of c->att (row stride D + 64), the decoder writes e beside it and
// turns c->mr into the statistics of x + a; the block's QKV GEMM then takes [x16 | e] x [W'_qkv | W'_qkv Q]
int run_adapter(hg_ctx* c, const AdapterW& a, int n_seq, int L, int D, const AdapterCall& ac, hipStream_t s, bool fused,
                int kcat = 0, half_t* e2 = nullptr);

// LayerNorm folded into the GEMMs: the residual GEMMs (out-proj, c_proj) also emit the fp16 copy of the updated
// rows and per-row partial statistics; the consuming GEMMs (QKV, c_fc) read that copy and apply mean / rstd in
// their epilogue.  Used when every GEMM of the block is eligible for the ring kernels and no adapter rewrites
// the stream between the residual GEMM and its LayerNorm.  Option ln_fuse = 0 selects the separate-LayerNorm path.
bool ln_fuse_ok(hg_ctx* c, int M, int D) {
    // Whenever the shapes are eligible (M >= 512): one arithmetic for every batch size keeps a row's result
    // independent of the batch it is in.  Option ln_fuse = 0 selects the separate-LayerNorm path.
    if (!c->opt_ln_fuse) return false;
    if (D % 256) return false;
    GemmArgs g{};
    float dummy = 0.f;
    g.cs = &dummy; g.mr = &dummy; g.M = M; g.K = D; g.lda = D;
    g.N = 3 * D; g.ldc = 3 * D;
    if (!gemm_ln_ok(EPI_LN_BIAS_F16, g)) return false;
    g.N = 4 * D; g.ldc = 4 * D;
    if (!gemm_ln_ok(EPI_LN_BIAS_QGELU_F16, g)) return false;
    GemmArgs r{};
    r.out2 = (half_t*)&dummy; r.stats = &dummy; r.mu = &dummy; r.stats_ld = 4 * (D / 256); r.M = M; r.N = D; r.ldc = D;
    r.K = D; r.lda = D;
    if (!gemm_ln_ok(EPI_RESID_LN_F32, r)) return false;
    r.K = 4 * D; r.lda = 4 * D;
    return gemm_ln_ok(EPI_RESID_LN_F32, r);
}

// `row0_out` (towers without token outputs): only ONE row of every sequence leaves the tower - row sel[seq] (the EOT
// token of the text tower) or row 0 when sel is null (the class token of the vision tower) - so the LAST block
// computes K and V for all rows but Q, attention, out-proj and the MLP for those n_seq rows only (a dense
// [n_seq, D] stream, returned through *row0_out); all other rows of that block never reach any output.
// Option last_block_row0 = 0 runs the last block on every row like the others.
// `pre_w` / `pre_b` (vision tower): the LayerNorm in front of the first block (ln_pre) has NOT been applied yet; it runs
// here, fused with the first block's folding statistics when folding is on - and, with `pre_pos` / `pre_cls`, with the class
// rows and the positional embedding the patch GEMM left out (clipnet/model.py:223-225 in one pass over the rows)
int run_blocks(hg_ctx* c, const std::vector<BlockW>& blocks, int n_seq, int L, int D, int heads, bool causal,
               hipStream_t s, float* trace, int trace_stride, const AdapterCall* ac, bool ln_fold,
               const float** row0_out = nullptr, const int32_t* sel = nullptr, const float* pre_w = nullptr,
               const float* pre_b = nullptr, const float* pre_pos = nullptr, const float* pre_cls = nullptr, bool gamma_act = false) {
    const int M = n_seq * L;
    const bool row0_env = c->opt_row0 != 0;
    if (row0_out) *row0_out = nullptr;
    if (c->pair_err && *(volatile int32_t*)c->pair_err) {
        *(volatile int32_t*)c->pair_err = 0;
        if (c->range_flag) *(volatile int32_t*)c->range_flag = 0;      // (whatever that call's rows overflowed into is part of the same report)
        return fail(c, HG_ERR_HIP, "a hand-off wait inside the MLP pair kernel of a previous call timed out (a workgroup of its grid never "
                                   "became resident): that call's outputs are invalid; set option mlp_pair = 0 if this device cannot "
                                   "hold one workgroup per compute unit");
    }
    if (c->range_flag && *(volatile int32_t*)c->range_flag) {
        *(volatile int32_t*)c->range_flag = 0;
        return fail(c, HG_ERR_INVALID, "in a previous call activations left the fp16 range inside a tower: a row reached more than 65504 from its "
                                       "centre, which the centred fp16 copy the LayerNorm-folded GEMMs read cannot hold (with option stream_hilo "
                                       "the residual stream itself is held in it) - that call's embeddings of those rows are invalid.  Option "
                                       "ln_fuse = 0 selects the separate-LayerNorm path");
    }
    float* x = (float*)c->x.p;
    half_t* h = (half_t*)c->h.p;
    half_t* qkv = (half_t*)c->qkv.p;
    half_t* att = (half_t*)c->att.p;
    half_t* fc = (half_t*)c->fc.p;
    const bool adapters = ac && ac->enabled;
    // With adapters the folding survives when the adapter's up_proj can re-emit the fp16 copy and statistics
    // (EPI_SCALE_RESID_LN_F32, duo kernel): option adapter_fuse = 0 selects the separate path (fp32 -> fp16 copy of the stream
    // per adapter, two LayerNorm kernels per block)
    const bool adapter_fuse_on = c->opt_adapter_fuse != 0;
    bool fuse = ln_fold && ln_fuse_ok(c, M, D);
    if (fuse && adapters) {
        GemmArgs u{};
        float dummy = 0.f;
        u.out2 = (half_t*)&dummy; u.stats = &dummy; u.mu = &dummy; u.pos = &dummy; u.stats_ld = 4 * (D / 256);
        u.M = M; u.N = D; u.ldc = D; u.K = 64; u.lda = 64;
        fuse = adapter_fuse_on && gemm_duo_ok(EPI_SCALE_RESID_LN_F32, u);
    }
    // How the adapter of block i reaches the stream:
    //   2  (default, option adapter_fold = 1, when the shapes allow) folded into the block's own GEMMs: QKV takes [x16 | e], out-proj
    //      [att | e], ln_1's statistics come from the decoder kernel (run_adapter) - no up_proj launch at all
    //   0  up_proj GEMM with a scaled-residual epilogue on the fp32 stream (+ fp16 copy + statistics when folding LayerNorm)
    const int kcat_env = c->opt_adapter_fold ? 2 : 0;
    std::vector<int> kmode(blocks.size(), 0);
    bool any_k2 = false;
    if (adapters && fuse)
        for (size_t i = 0; i < blocks.size(); ++i) {
            if (c->vit.adapters.size() <= i || !c->vit.adapters[i].present) continue;
            const AdapterW& aw = c->vit.adapters[i];
            if (row0_out && row0_env && i + 1 == blocks.size()) continue;
            if (kcat_env >= 2 && aw.fold[ac->priors ? 0 : 1].wq_cat) {
                AdapterDev ad{};
                for (int k = 0; k < 2; ++k) ad.w16[k][0] = aw.w16[k][0];
                GemmArgs q{};
                float dummy = 0.f;
                q.cs = &dummy; q.mr = &dummy; q.M = M; q.K = D + 64; q.lda = D + 64; q.N = 3 * D; q.ldc = 3 * D;
                if (adapter_decoder_mfma_ok(ad, ac->priors != nullptr, L, ac->N) && gemm_ln_ok(EPI_LN_BIAS_F16, q)) {
                    kmode[i] = 2;
                    any_k2 = true;
                }
            }
        }
    if (any_k2) {
        int rc = ensure(c, c->att, rup(M, 256) * (size_t)(D + 64) * 2);
        if (rc) return rc;
        att = (half_t*)c->att.p;
    }
    // the centred fp16 copy a block's ln_1 reads: beside e in the out-proj operand buffer when its adapter is folded (mode 2)
    auto h_of = [&](size_t i) { return i < blocks.size() && kmode[i] == 2 ? att : h; };
    auto ldh_of = [&](size_t i) { return i < blocks.size() && kmode[i] == 2 ? D + 64 : D; };
    const int sld = 4 * (D / 256);
    float* mr = nullptr;
    float* mu = nullptr;
    float* muc = nullptr;
    float* stats = nullptr;
    if (fuse) {
        int rc = ensure(c, c->mr, rup(M, 256) * 2 * 4);
        if (!rc) rc = ensure(c, c->mu, rup(M, 256) * 4);
        if (!rc) rc = ensure(c, c->stats, rup(M, 256) * (size_t)sld * 2 * 4);
        if (rc) return rc;
        if (!rc && adapters) rc = ensure(c, c->muc, rup(M, 256) * 4);
        if (rc) return rc;
        mr = (float*)c->mr.p;
        mu = (float*)c->mu.p;
        muc = adapters ? (float*)c->muc.p : nullptr;
        stats = (float*)c->stats.p;
        if (pre_w) HG_HIP(launch_layernorm_rowstats(x, pre_w, pre_b, h_of(0), mr, mu, muc, M, D, s, pre_pos, pre_cls, L, ldh_of(0)));
        else {
            if (kmode.size() && kmode[0] == 2) kmode[0] = 0;      // rowstats_cast writes dense rows
            if (gamma_act && !adapters) {
                int rc2 = ensure(c, c->hg, rup(M, 256) * (size_t)D * 2);
                if (rc2) return rc2;
                HG_HIP(launch_rowstats_cast(x, (half_t*)c->hg.p, mr, mu, M, D, s, muc, blocks[0].ln1_w));
            } else {
                HG_HIP(launch_rowstats_cast(x, h, mr, mu, M, D, s, muc));
            }
        }
    } else if (pre_w) {
        HG_HIP(launch_layernorm_f32(x, pre_w, pre_b, x, M, D, s, pre_pos, pre_cls, L));
    }
    if (pre_w && trace) HG_HIP(launch_copy_rows(x, trace, n_seq, L, D, s));
    // Residual stream as centre + hi + lo between the LayerNorm-emitting residual GEMMs (GemmArgs::hl; option stream_hilo):
    // hi IS the centred fp16 copy those GEMMs write anyway, lo its remainder as bf8 (HG_LO8; fp16 otherwise) - 6 (8) instead of 10
    // bytes per element through every such epilogue and a third fewer partial-line stores.  The first of them reads the fp32 stream (ln_pre wrote it),
    // the last one writes fp32 again (the plain last c_proj / the class-rows path / ln_post read it); nothing in between
    // touches x.  Variant A, and variant C when EVERY block's adapter is folded into its GEMMs (mode 2: nothing but the residual GEMMs
    // rewrites the stream; round 5): the hi half then lives in the in_proj operand buffer [x16 | e] (row stride D + 64) through the
    // whole block - attention writes into a second operand buffer [att | e] instead of over it, the decoder writes e into both, c_fc
    // reads the copy with that stride.  Only where the GEMMs run on gemm_ring2.
    const bool row0_plan = row0_out && row0_env && !adapters;
    const int n_rln = fuse ? (row0_plan ? 2 * ((int)blocks.size() - 1) : 2 * (int)blocks.size() - 1) : 0;
    bool all_k2 = adapters && !kmode.empty();
    for (int k : kmode) all_k2 = all_k2 && k == 2;
    // (a per-block trace of variant C reads the stream's hi half with row stride D: variant C's lives in [x16 | e] with stride D + 64 - no
    // entry point asks for that trace today; should one, it gets the fp32 stream)
    bool hilo = fuse && (!adapters || all_k2) && c->opt_stream_hilo && n_rln >= 2 && !(trace && adapters);
    if (hilo) {
        GemmArgs r{};
        r.M = M; r.N = D; r.ldc = D; r.K = adapters ? D + 64 : D; r.lda = r.K;
        hilo = gemm_ring2_ok(r);
        r.K = 4 * D; r.lda = 4 * D;
        hilo = hilo && gemm_ring2_ok(r);
    }
    half_t* att2 = nullptr;
    if (hilo) {
        int rc = ensure(c, c->xlo, gemm_lo_bytes(M, D));
        if (!rc) rc = ensure(c, c->muc, rup(M, 256) * 4);
        if (!rc && adapters) rc = ensure(c, c->att2, rup(M, 256) * (size_t)(D + 64) * 2);
        if (rc) return rc;
        muc = (float*)c->muc.p;
        att2 = adapters ? (half_t*)c->att2.p : nullptr;
    }
    const bool hilo_c = hilo && adapters;      // (variant C on the hi / lo stream: att holds [x16 = hi | e], att2 [attention output | e])
    // blocks whose in_proj and attention run as one kernel (option qkv_attn): folded LayerNorm, no adapter in the block's
    // GEMMs, a sequence per row tile (192 < L <= 208), enough sequences to fill the chip
    const bool qa_on = fuse && !causal && c->opt_qkv_attn && n_seq >= c->opt_qkv_attn_min_seq && qkv_attn_ok(n_seq, L, D, heads, D) &&
                       (c->opt_qkv_attn == 2 || qkv_attn_pays(n_seq, heads, 0));
    // (variant C on the hi / lo stream: the same kernel over K = D + 64, [x16 | e] in att -> att2; without the second buffer the
    // attention output overwrites x16 in place, which other work items of the sequence are still reading)
    auto qa_block = [&](size_t i, bool row0_last_blk) {
        if (!qa_on || row0_last_blk || blocks[i].wp_qkv == nullptr) return false;
        if (kmode[i] == 0) return true;
        return kmode[i] == 2 && hilo_c && c->opt_qkv_attn_c && c->vit.adapters[i].fold[ac->priors ? 0 : 1].wp_qcat != nullptr &&
               qkv_attn_ok(n_seq, L, D, heads, D + 64, D + 64);
    };
    // gs (text tower): the LayerNorm weight rides in the activation copy (GemmArgs::gamma), the consumers multiply by the layer's own
    // fp16 weights: no second rounding of W * gamma.  a_ln = where the next consumer finds that copy: c->hg behind a producer that keeps
    // h as the stream's hi half (hl 1, 2), h itself otherwise
    const bool gs = gamma_act && fuse && !adapters && pre_w == nullptr;
    half_t* const hg = gs ? (half_t*)c->hg.p : nullptr;
    half_t* a_ln = gs ? hg : h;
    auto gs_args = [&](GemmArgs& g, const float* gamma) {
        if (!gs) return;
        g.gamma = gamma;
        if (g.hl == 1 || g.hl == 2) { g.out3 = hg; g.ld3 = D; a_ln = hg; }
        else a_ln = h;
    };
    // c_fc and c_proj of a block as ONE persistent launch (option mlp_pair, hg_mlp_pair.hip): LayerNorm-folded blocks without adapters
    // where c_proj is a LayerNorm-emitting residual GEMM on the hi / lo (or fp32) stream (every block but the tower's last)
    const int pair_panels = (int)rup(mlp_pair_ready_words(M), 64);      // words per block: the census, then a counter per 256-row panel
    const bool pair_on = fuse && !adapters && c->opt_mlp_pair && c->pair_err && blocks.size() > 1;
    if (pair_on) {
        int rc = ensure(c, c->pair_ready, blocks.size() * (size_t)pair_panels * 4);
        if (rc) return rc;
        HG_HIP(hipMemsetAsync(c->pair_ready.p, 0, blocks.size() * (size_t)pair_panels * 4, s));
    }
    int rln_i = 0;                  // index of the next LayerNorm-emitting residual GEMM
    bool x_is_hilo = false;         // the stream currently lives in (h, xlo, muc), not in x
    auto rln_args = [&](GemmArgs& g) {
        if (!hilo) return;
        g.hl = rln_i == 0 ? 1 : (rln_i == n_rln - 1 ? 3 : 2);
        g.lo = (half_t*)c->xlo.p; g.muc = muc;
        x_is_hilo = g.hl != 3;
        ++rln_i;
    };
    for (size_t i = 0; i < blocks.size(); ++i) {
        const BlockW& b = blocks[i];
        const int kcat = kmode[i];
        if (adapters && c->vit.adapters.size() > i && c->vit.adapters[i].present) {
            const AdapterW& aw = c->vit.adapters[i];
            if (kcat) {
                int rc = ensure(c, c->att, rup(M, 256) * (size_t)(D + 64) * 2);      // (mode 2: sized before the loop)
                if (rc) return rc;
                att = (half_t*)c->att.p;
            }
            int rc = run_adapter(c, aw, n_seq, L, D, *ac, s, fuse, kcat, hilo_c ? att2 + D : nullptr);
            if (rc) return rc;
        }
        const bool row0_last = row0_out && row0_env && !adapters && i + 1 == blocks.size();
        // in_proj rows [qoff, 3D): the class-rows-only last block needs K and V of every row but Q of row 0 only
        const size_t qoff = row0_last ? D : 0;
        GemmArgs g{};
        g.A = h; g.lda = D; g.out = qkv + qoff; g.ldc = 3 * D; g.M = M; g.N = 3 * D - (int)qoff; g.K = D;
        if (fuse && kcat == 2 && qa_block(i, row0_last)) {
            QkvAttnArgs qa{};
            qa.x16 = att; qa.lda = D + 64; qa.K = D + 64; qa.wp = c->vit.adapters[i].fold[ac->priors ? 0 : 1].wp_qcat; qa.bcs = b.bcs_qkv;
            qa.mr = mr; qa.out = att2; qa.ldo = D + 64;
            qa.n_seq = n_seq; qa.L = L; qa.D = D; qa.heads = heads; qa.gsz = c->opt_qkv_attn_gsz;
            qa.a_bytes = (unsigned)(rup(M, 256) * (size_t)(D + 64) * 2);
            ProfScope ps(c, s, HG_PROF_QKV_ATTN, n_seq, L, heads);
            HG_HIP(launch_qkv_attn(qa, s));
        } else if (fuse && kcat == 2) {      // ln_1(x + a) W^T: [x16 | e] x [W'_qkv | W'_qkv Q], the statistics are those of x + a
            g.A = att; g.lda = D + 64; g.K = D + 64;
            g.W = c->vit.adapters[i].fold[ac->priors ? 0 : 1].wq_cat; g.bias = b.bf_qkv; g.cs = b.cs_qkv; g.mr = mr;
            HG_HIP(gemm(c, EPI_LN_BIAS_F16, g, s));
        } else if (fuse && qa_block(i, row0_last)) {
            // in_proj + attention in one kernel: q, k, v stay in LDS (hg_qkv_attn.hip); bit-identical to the two kernels below
            QkvAttnArgs qa{};
            qa.x16 = h; qa.lda = D; qa.wp = b.wp_qkv; qa.bcs = b.bcs_qkv; qa.mr = mr; qa.out = att; qa.ldo = D;
            qa.n_seq = n_seq; qa.L = L; qa.D = D; qa.heads = heads; qa.gsz = c->opt_qkv_attn_gsz;
            qa.a_bytes = (unsigned)(rup(M, 256) * (size_t)D * 2);
            ProfScope ps(c, s, HG_PROF_QKV_ATTN, n_seq, L, heads);
            HG_HIP(launch_qkv_attn(qa, s));
        } else if (fuse) {
            g.W = b.wf_qkv + qoff * D; g.bias = b.bf_qkv + qoff; g.cs = b.cs_qkv + qoff; g.mr = mr;
            if (gs) { g.A = a_ln; g.W = b.w_qkv + qoff * D; g.cs = b.csg_qkv + qoff; }
            HG_HIP(gemm(c, EPI_LN_BIAS_F16, g, s));
        } else {
            HG_HIP(launch_layernorm_f16(x, b.ln1_w, b.ln1_b, h, M, D, nullptr, 0, 1, s));
            g.W = b.w_qkv + qoff * D; g.bias = b.b_qkv + qoff;
            HG_HIP(gemm(c, EPI_BIAS_F16, g, s));
        }
        if (row0_last) {
            int rc = ensure(c, c->cx, (size_t)n_seq * D * 4);
            if (!rc) rc = ensure(c, c->ca, rup(n_seq, 256) * D * 2);
            if (!rc) rc = ensure(c, c->ch, rup(n_seq, 256) * D * 2);
            if (!rc) rc = ensure(c, c->cf, rup(n_seq, 256) * (size_t)4 * D * 2);
            if (!rc) rc = ensure(c, c->cq, rup(n_seq, 256) * D * 2);
            if (rc) return rc;
            float* cx = (float*)c->cx.p;
            half_t *ca = (half_t*)c->ca.p, *ch = (half_t*)c->ch.p, *cf = (half_t*)c->cf.p, *cq = (half_t*)c->cq.p;
            // Q of the selected rows: ln_1 on those rows, then the first D rows of in_proj
            if (sel) HG_HIP(launch_layernorm_f16(x, b.ln1_w, b.ln1_b, ch, n_seq, D, sel, L, 0, s));
            else HG_HIP(launch_layernorm_f16(x, b.ln1_w, b.ln1_b, ch, n_seq, D, nullptr, 0, L, s));
            g = GemmArgs{};
            g.A = ch; g.lda = D; g.W = b.w_qkv; g.bias = b.b_qkv; g.out = cq; g.ldc = D; g.M = n_seq; g.N = D; g.K = D;
            HG_HIP(gemm(c, EPI_BIAS_F16, g, s));
            HG_HIP(launch_attention_row0(qkv, cq, sel, ca, n_seq, L, heads, causal, s));
            HG_HIP(launch_copy_rows(x, cx, n_seq, L, D, s, sel));
            g = GemmArgs{};
            g.A = ca; g.lda = D; g.W = b.w_out; g.bias = b.b_out; g.out = cx; g.ldc = D; g.M = n_seq; g.N = D; g.K = D;
            HG_HIP(gemm(c, EPI_BIAS_RESID_F32, g, s));
            HG_HIP(launch_layernorm_f16(cx, b.ln2_w, b.ln2_b, ch, n_seq, D, nullptr, 0, 1, s));
            g = GemmArgs{};
            g.A = ch; g.lda = D; g.W = b.w_fc; g.bias = b.b_fc; g.out = cf; g.ldc = 4 * D; g.M = n_seq; g.N = 4 * D; g.K = D;
            HG_HIP(gemm(c, EPI_BIAS_QGELU_F16, g, s));
            g = GemmArgs{};
            g.A = cf; g.lda = 4 * D; g.W = b.w_proj; g.bias = b.b_proj; g.out = cx; g.ldc = D; g.M = n_seq; g.N = D; g.K = 4 * D;
            HG_HIP(gemm(c, EPI_BIAS_RESID_F32, g, s));
            if (trace) HG_HIP(launch_copy_rows(cx, trace + (size_t)(i + 1) * trace_stride, n_seq, 1, D, s));
            *row0_out = cx;
            break;
        }
        half_t* const att_o = hilo_c ? att2 : att;      // where the attention output (the out-proj operand) goes
        if (!(fuse && qa_block(i, row0_last))) HG_HIP(attention(c, qkv, att_o, n_seq, L, heads, causal, s, kcat ? D + 64 : 0));
        g = GemmArgs{};
        g.A = att_o; g.lda = D; g.W = b.w_out; g.bias = b.b_out; g.out = x; g.ldc = D; g.M = M; g.N = D; g.K = D;
        if (kcat == 2) {      // x += [att | e] [W_out | Q]^T + b_out: the adapter's update rides along
            g.lda = D + 64; g.K = D + 64; g.W = c->vit.adapters[i].fold[ac->priors ? 0 : 1].wk_out;
        }
        if (fuse) {
            g.out2 = h; g.stats = stats; g.stats_ld = sld; g.mu = mu;
            if (hilo_c) { g.out2 = att; g.ld2 = D + 64; }      // the copy = the stream's hi half stays in [x16 | e]
            if (!kcat || hilo_c) rln_args(g);
            gs_args(g, b.ln2_w);
            HG_HIP(gemm(c, EPI_RESID_LN_F32, g, s));
            HG_HIP(launch_finalize_stats(stats, mr, mu, M, sld, 64, s, muc, false, c->range_flag));
        } else {
            HG_HIP(gemm(c, EPI_BIAS_RESID_F32, g, s));
        }
        g = GemmArgs{};
        g.A = h; g.lda = D; g.out = fc; g.ldc = 4 * D; g.M = M; g.N = 4 * D; g.K = D;
        int mlp_done = 0;      // leading rows whose MLP ran as the one kernel (separate-LayerNorm path, width 512)
        GemmArgs pq{};         // option mlp_pair: the c_proj arguments, built ahead of c_fc (pq_built), and whether the pair kernel took both
        bool pq_built = false, paired = false, pair_fin = false;
        if (fuse) {
            g.W = b.wf_fc; g.bias = b.bf_fc; g.cs = b.cs_fc; g.mr = mr;
            if (hilo_c) { g.A = att; g.lda = D + 64; }
            if (gs) { g.A = a_ln; g.W = b.w_fc; g.cs = b.csg_fc; }
            if (pair_on && i + 1 < blocks.size()) {
                pq.A = fc; pq.lda = 4 * D; pq.W = b.w_proj; pq.bias = b.b_proj; pq.out = x; pq.ldc = D; pq.M = M; pq.N = D; pq.K = 4 * D;
                pq.out2 = h_of(i + 1); pq.ld2 = ldh_of(i + 1); pq.stats = stats; pq.stats_ld = sld; pq.mu = mu;
                rln_args(pq);
                gs_args(pq, blocks[i + 1].ln1_w);
                pq_built = true;
                if (mlp_pair_ok(g, pq, c->n_cu)) {
                    PairGateScope gate(c->device, s);
                    ProfScope ps(c, s, HG_PROF_MLP_PAIR, M, 4 * D, D);
                    // (option mlp_pair = 2: the launch's tail also does finalize_stats' work - measured +10 us on the launch for the 5 us
                    // launch it removes in the vision tower, a tie in the text tower (profiles/r06_mlp_pair.txt item 8); 1: its own launch)
                    const bool fin = c->opt_mlp_pair == 2;
                    HG_HIP(launch_mlp_pair(g, pq, (unsigned*)c->pair_ready.p + i * (size_t)pair_panels, c->pair_err,
                                           c->opt_mlp_pair_chunk, c->opt_mlp_pair_fc_slots, c->n_cu, s, fin ? mr : nullptr, mu, muc,
                                           c->range_flag, c->opt_mlp_pair_fault));
                    paired = true;
                    pair_fin = fin;
                }
            }
            if (!paired) HG_HIP(gemm(c, EPI_LN_BIAS_QGELU_F16, g, s));
        } else {
            HG_HIP(launch_layernorm_f16(x, b.ln2_w, b.ln2_b, h, M, D, nullptr, 0, 1, s));
            // Width 512 (the text tower): x += W_proj quickgelu(W_fc h + b_fc) + b_proj as ONE kernel for the leading rows that fill
            // whole rounds of its 128-row items (hg_vae_fused.hip mode 3: the [rows, 2048] activation stays on chip); the rest below
            mlp_done = (b.wp_mlp && !trace) ? fused_item_rows(c, c->opt_mlp_fused, M) : 0;
            if (mlp_done > 0) {
                VaeFusedArgs a{};
                a.x16 = h; a.bias = x; a.wp = b.wp_mlp; a.b0g = b.b_fc; a.b2g = b.b_proj;
                a.R = mlp_done; a.eh = 0; a.gh = 4 * D; a.mode = 3; a.has_enc = false;
                ProfScope ps(c, s, HG_PROF_VAE_FUSED, mlp_done, 1, 4 * D);
                HG_HIP(launch_vae_fused(a, s));
            }
            if (mlp_done < M) {
                g.A = h + (size_t)mlp_done * D; g.M = M - mlp_done;
                g.W = b.w_fc; g.bias = b.b_fc;
                HG_HIP(gemm(c, EPI_BIAS_QGELU_F16, g, s));
            }
        }
        if (mlp_done >= M) continue;      // (only without a trace, on the separate-LayerNorm path: the stream is the fp32 x)
        g = GemmArgs{};
        g.A = fc; g.lda = 4 * D; g.W = b.w_proj; g.bias = b.b_proj; g.out = x + (size_t)mlp_done * D; g.ldc = D; g.M = M - mlp_done; g.N = D;
        g.K = 4 * D;
        if (fuse && i + 1 < blocks.size()) {      // the last block is followed by ln_post / ln_final on selected rows
            if (pq_built) {
                g = pq;
            } else {
                g.out2 = h_of(i + 1); g.ld2 = ldh_of(i + 1); g.stats = stats; g.stats_ld = sld; g.mu = mu;
                rln_args(g);
                gs_args(g, blocks[i + 1].ln1_w);
            }
            if (!paired) HG_HIP(gemm(c, EPI_RESID_LN_F32, g, s));
            if (!pair_fin) HG_HIP(launch_finalize_stats(stats, mr, mu, M, sld, 64, s, muc, false, c->range_flag));
        } else {
            HG_HIP(gemm(c, EPI_BIAS_RESID_F32, g, s));
        }
        if (trace) {
            // (after finalize_stats muc is the centre the stream's hi / lo halves were written with)
            if (x_is_hilo) HG_HIP(launch_copy_rows_hilo(h, (const half_t*)c->xlo.p, muc, trace + (size_t)(i + 1) * trace_stride, n_seq, L, D, s));
            else HG_HIP(launch_copy_rows(x, trace + (size_t)(i + 1) * trace_stride, n_seq, L, D, s));
        }
    }
    return HG_OK;
}

int ensure_tower_ws(hg_ctx* c, int M, int D) {
    const size_t Mp = rup(M, 256);
    int rc = 0;
    keep_first(rc, ensure(c, c->x, Mp * D * 4));
    keep_first(rc, ensure(c, c->h, Mp * D * 2));
    keep_first(rc, ensure(c, c->qkv, Mp * 3 * D * 2));
    keep_first(rc, ensure(c, c->att, Mp * D * 2));
    keep_first(rc, ensure(c, c->fc, Mp * 4 * D * 2));
    return rc;
}

int run_adapter(hg_ctx* c, const AdapterW& a, int n_seq, int L, int D, const AdapterCall& ac, hipStream_t s, bool fused,
                int kcat, half_t* e2) {
    const int M = n_seq * L;
    float* x = (float*)c->x.p;
    half_t* h = (half_t*)c->h.p;
    const size_t Mp = rup(M, 256);
    int rc = ensure(c, c->ad32, Mp * 128 * 4);
    if (!rc) rc = ensure(c, c->ad16, Mp * 64 * 2);
    const int Nmem = ac.priors ? ac.N : L;
    if (!rc) rc = ensure(c, c->adkv, (size_t)n_seq * Nmem * 64 * 4 * 2);
    if (rc) return rc;
    AdapterDev ad{};
    ad.down_w = a.down_w; ad.down_b = a.down_b; ad.up_w = a.up_w; ad.up_b = a.up_b; ad.scale = a.scale;
    for (int k = 0; k < 2; ++k) {
        for (int j = 0; j < 12; ++j) ad.dl[k][j] = a.dl[k][j];
        for (int j = 0; j < 6; ++j) ad.w16[k][j] = a.w16[k][j];
    }
    // down = relu(down_proj(x))  (CLIP_models_adapter_prior2.py:184-185) - inside the decoder kernel when it can (folded mode)
    const bool down_fused = kcat == 2 && adapter_decoder_fused_down_ok(ad, ac.priors != nullptr, L, ac.N);
    const AdapterW::Fold& fold = a.fold[ac.priors ? 0 : 1];
    GemmArgs g{};
    if (!down_fused) {
        g.A = h; g.lda = D; g.W = a.down_w; g.bias = a.down_b; g.out = c->ad32.p; g.ldc = 128; g.M = M; g.N = 128; g.K = D;
        if (kcat == 2) {   // ... read from the out-proj operand buffer, with x16 Q in the padded columns (no ReLU there)
            g.A = (const half_t*)c->att.p; g.lda = D + 64; g.W = fold.down2; g.n_split = 64;
            g.cs = a.down_cs; g.mu = (const float*)c->muc.p;
            HG_HIP(gemm(c, EPI_MU_BIAS_RELU_F32, g, s));
        } else if (fused) {       // on the centred fp16 copy the residual GEMMs keep current: W (x16 + mu) + b
            g.cs = a.down_cs; g.mu = (const float*)c->muc.p;
            HG_HIP(gemm(c, EPI_MU_BIAS_RELU_F32, g, s));
        } else {
            HG_HIP(launch_f32_to_f16(x, h, (size_t)M * D, s));
            HG_HIP(gemm(c, EPI_BIAS_RELU_F32, g, s));
        }
    }
    // post-norm decoder layer(s) over the 64-wide bottleneck (adapter...:186-200); with adapter_num_layers > 1 the
    // prior path chains mhsa_layers.0 .. N-1, the intermediate activations staying fp32 in place
    const int n_chain = ac.priors ? 1 + (int)a.extra.size() : 1;
    // chained layers exist only in the MFMA decoder (one workgroup per sequence, <= 32 prior tokens): say so up front
    // instead of failing inside the launch
    if (n_chain > 1 && !adapter_decoder_mfma_ok(ad, true, L, ac.N))
        return fail(c, HG_ERR_INVALID, "adapter_num_layers > 1 needs the MFMA decoder path: at most 32 prior tokens (got %d)", ac.N);
    for (int z = 0; z < n_chain; ++z) {
        if (z > 0)
            for (int j = 0; j < 12; ++j) ad.dl[0][j] = a.extra[z - 1].dl[j];
        if (z > 0)
            for (int j = 0; j < 6; ++j) ad.w16[0][j] = a.extra[z - 1].w16[j];
        half_t* d16 = kcat ? (half_t*)c->att.p + D : (half_t*)c->ad16.p;
        AdapterFoldDev fd{};
        if (kcat == 2) { fd.g16 = fold.g16; fd.qm = fold.qm; fd.mr = (float*)c->mr.p; fd.inv_D = 1.0f / (float)D; fd.e2 = e2; }
        AdapterDownDev dn{};
        if (down_fused && z == 0) {
            dn.x16 = (const half_t*)c->att.p; dn.ldx = D + 64; dn.K = D; dn.w = fold.down2; dn.b = a.down_b; dn.cs = a.down_cs;
            dn.muc = (const float*)c->muc.p;
        }
        hipError_t e = launch_adapter_decoder((const float*)c->ad32.p, ad, ac.priors, ac.mask, n_seq, L, ac.priors ? ac.N : 0,
                                              (float*)c->adkv.p, d16, s,
                                              z + 1 < n_chain ? (float*)c->ad32.p : nullptr, kcat ? D + 64 : 64,
                                              kcat == 2 ? &fd : nullptr, down_fused && z == 0 ? &dn : nullptr);
        if (e != hipSuccess)
            return fail(c, HG_ERR_HIP, "adapter decoder layer %d failed: %s", z, hipGetErrorString(e));
    }
    if (kcat == 2) return HG_OK;      // the update itself happens in the block's QKV and out-proj GEMMs
    // x += up_proj(.) * scale   (adapter...:201-202, :456)
    g = GemmArgs{};
    g.A = (const half_t*)c->ad16.p; g.lda = 64; g.W = a.up_w; g.bias = a.up_b; g.pos = a.scale; g.out = x; g.ldc = D;
    g.M = M; g.N = D; g.K = 64;
    if (fused) {       // ... and re-emit the fp16 copy + row statistics of the updated stream for the folded ln_1
        const int sld = 4 * (D / 256);
        g.out2 = h; g.stats = (float*)c->stats.p; g.stats_ld = sld; g.mu = (const float*)c->mu.p;
        HG_HIP(gemm(c, EPI_SCALE_RESID_LN_F32, g, s));
        HG_HIP(launch_finalize_stats((const float*)c->stats.p, (float*)c->mr.p, (float*)c->mu.p, M, sld, 64, s,
                                     (float*)c->muc.p));
    } else {
        HG_HIP(gemm(c, EPI_SCALE_RESID_F32, g, s));
    }
    return HG_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* hg_version(void) { return "hoigen_amd 0.1 (gfx950)"; }

hg_ctx* hg_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return nullptr;      // creates the primary context if needed
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    hg_ctx* c = new hg_ctx();
    c->device = device;
    {
        DevGuard g(c);
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) c->n_cu = ncu;
        if (hipHostMalloc((void**)&c->eot_flag, 64, hipHostMallocMapped) == hipSuccess && c->eot_flag) *c->eot_flag = 0;
        else c->eot_flag = nullptr;
        if (hipMalloc((void**)&c->eot_flag_dev, 64) != hipSuccess) c->eot_flag_dev = nullptr;
        else (void)hipMemset(c->eot_flag_dev, 0, 64);
        if (hipHostMalloc((void**)&c->pair_err, 64, hipHostMallocMapped) == hipSuccess && c->pair_err) *c->pair_err = 0;
        else c->pair_err = nullptr;
        if (hipHostMalloc((void**)&c->range_flag, 64, hipHostMallocMapped) == hipSuccess && c->range_flag) *c->range_flag = 0;
        else c->range_flag = nullptr;
    }
    struct { const char* env; const char* key; } init[] = {{"HG_CHUNK_ROWS", "chunk_rows"}, {"HG_LAST_BLOCK_ROW0", "last_block_row0"},
                                                           {"HG_LN_FUSE", "ln_fuse"}, {"HG_ADAPTER_FUSE", "adapter_fuse"},
                                                           {"HG_ADAPTER_FOLD", "adapter_fold"}, {"HG_STREAM_HILO", "stream_hilo"},
                                                           {"HG_QKV_ATTN", "qkv_attn"}, {"HG_QKV_ATTN_MIN_SEQ", "qkv_attn_min_seq"},
                                                           {"HG_QKV_ATTN_GSZ", "qkv_attn_gsz"}, {"HG_QKV_ATTN_C", "qkv_attn_c"}, {"HG_TEXT_LN_FOLD", "text_ln_fold"}, {"HG_VAE_FUSED", "vae_fused"}, {"HG_MLP_FUSED", "mlp_fused"},
                                                           {"HG_MLP_PAIR", "mlp_pair"},
                                                           {"HG_MLP_PAIR_CHUNK", "mlp_pair_chunk"}, {"HG_MLP_PAIR_FC_SLOTS", "mlp_pair_fc_slots"},
                                                           {"HG_MLP_PAIR_FAULT", "mlp_pair_fault"}};
    for (auto& o : init)
        if (const char* e = getenv(o.env)) (void)hg_set_option(c, o.key, atoi(e));      // (out-of-range values are ignored)
    c->err.clear();
    return c;
}

int hg_set_option(hg_ctx* c, const char* key, int value) {
    if (!c || !key) return HG_ERR_INVALID;
    const std::string k(key);
    if (k == "chunk_rows") {          // rows per VAE / mlp_net / cache-logits chunk
        if (value < 256) return fail(c, HG_ERR_INVALID, "chunk_rows must be >= 256");
        c->max_chunk_rows = value;
    } else if (k == "last_block_row0") c->opt_row0 = value != 0;
    else if (k == "ln_fuse") c->opt_ln_fuse = value != 0;
    else if (k == "adapter_fuse") c->opt_adapter_fuse = value != 0;
    else if (k == "adapter_fold") c->opt_adapter_fold = value != 0;
    else if (k == "stream_hilo") c->opt_stream_hilo = value != 0;
    else if (k == "qkv_attn") {
        if (value < 0 || value > 2) return fail(c, HG_ERR_INVALID, "qkv_attn must be 0, 1 or 2 (got %d)", value);
        c->opt_qkv_attn = value;
    } else if (k == "qkv_attn_min_seq") {
        if (value < 1) return fail(c, HG_ERR_INVALID, "qkv_attn_min_seq must be >= 1 (got %d)", value);
        c->opt_qkv_attn_min_seq = value;
    } else if (k == "qkv_attn_gsz") {
        if (value < 0 || value > 6) return fail(c, HG_ERR_INVALID, "qkv_attn_gsz must be 0 .. 6 (got %d)", value);
        c->opt_qkv_attn_gsz = value;
    } else if (k == "text_ln_fold") {
        if (value < 0 || value > 2) return fail(c, HG_ERR_INVALID, "text_ln_fold must be 0, 1 or 2 (got %d)", value);
        c->opt_text_ln_fold = value;
    } else if (k == "qkv_attn_c") {
        if (value < 0 || value > 1) return fail(c, HG_ERR_INVALID, "qkv_attn_c must be 0 or 1 (got %d)", value);
        c->opt_qkv_attn_c = value;
    }
    else if (k == "vae_fused") {
        if (value < 0 || value > 2) return fail(c, HG_ERR_INVALID, "vae_fused must be 0, 1 or 2 (got %d)", value);
        c->opt_vae_fused = value;
    } else if (k == "mlp_fused") {
        if (value < 0 || value > 2) return fail(c, HG_ERR_INVALID, "mlp_fused must be 0, 1 or 2 (got %d)", value);
        c->opt_mlp_fused = value;
        if (value && c->text.loaded) {      // the packed operand of that kernel is made on demand
            HG_ON_DEVICE(c);
            int rc = pack_mlp_blocks(c, c->text.owned, c->text.blocks, c->text.D);
            if (rc) return rc;
        }
    } else if (k == "mlp_pair") {
        if (value < 0 || value > 2) return fail(c, HG_ERR_INVALID, "mlp_pair must be 0, 1 or 2 (got %d)", value);
        c->opt_mlp_pair = value;
    } else if (k == "mlp_pair_chunk") {
        if (value < 1 || value > 64) return fail(c, HG_ERR_INVALID, "mlp_pair_chunk must be 1 .. 64 (got %d)", value);
        c->opt_mlp_pair_chunk = value;
    } else if (k == "mlp_pair_fc_slots") {
        if (value < 1 || value > 64) return fail(c, HG_ERR_INVALID, "mlp_pair_fc_slots must be 1 .. 64 (got %d)", value);
        c->opt_mlp_pair_fc_slots = value;
    } else if (k == "mlp_pair_fault") {
        if (value < 0 || value > 1) return fail(c, HG_ERR_INVALID, "mlp_pair_fault must be 0 or 1 (got %d)", value);
        c->opt_mlp_pair_fault = value;
    }
    else return fail(c, HG_ERR_INVALID, "unknown option '%s'", key);
    return HG_OK;
}

int hg_get_option(hg_ctx* c, const char* key, int* value) {
    if (!c || !key || !value) return HG_ERR_INVALID;
    const std::string k(key);
    if (k == "chunk_rows") *value = c->max_chunk_rows;
    else if (k == "last_block_row0") *value = c->opt_row0;
    else if (k == "ln_fuse") *value = c->opt_ln_fuse;
    else if (k == "adapter_fuse") *value = c->opt_adapter_fuse;
    else if (k == "adapter_fold") *value = c->opt_adapter_fold;
    else if (k == "stream_hilo") *value = c->opt_stream_hilo;
    else if (k == "stream_lo_bits") *value = HG_LO8 ? 8 : 16;      // read-only: how the build holds the low half
    else if (k == "qkv_attn") *value = c->opt_qkv_attn;
    else if (k == "qkv_attn_min_seq") *value = c->opt_qkv_attn_min_seq;
    else if (k == "qkv_attn_gsz") *value = c->opt_qkv_attn_gsz;
    else if (k == "qkv_attn_c") *value = c->opt_qkv_attn_c;
    else if (k == "text_ln_fold") *value = c->opt_text_ln_fold;
    else if (k == "vae_fused") *value = c->opt_vae_fused;
    else if (k == "mlp_fused") *value = c->opt_mlp_fused;
    else if (k == "mlp_pair") *value = c->opt_mlp_pair;
    else if (k == "mlp_pair_chunk") *value = c->opt_mlp_pair_chunk;
    else if (k == "mlp_pair_fc_slots") *value = c->opt_mlp_pair_fc_slots;
    else if (k == "mlp_pair_fault") *value = c->opt_mlp_pair_fault;
    else return fail(c, HG_ERR_INVALID, "unknown option '%s'", key);
    return HG_OK;
}

void hg_destroy(hg_ctx* c) {
    if (!c) return;
    DevGuard dev_guard_(c);
    (void)hipDeviceSynchronize();
    free_all(c->vit.owned);
    free_all(c->vit.owned_adapters);
    free_all(c->text.owned);
    for (auto& v : c->vae) free_all(v.owned);
    for (auto& m : c->mlp) free_all(m.owned);
    for (auto& m : c->cache) free_all(m.owned);
    Buf* bufs[] = {&c->x, &c->h, &c->qkv, &c->att, &c->fc, &c->head16, &c->tok32, &c->small, &c->i32,
                   &c->ad32, &c->ad16, &c->adkv, &c->mr, &c->mu, &c->muc, &c->stats, &c->pre, &c->pretab, &c->cx, &c->ca, &c->ch, &c->cf, &c->cq, &c->xlo, &c->zpark, &c->att2, &c->hg, &c->pair_ready};
    for (Buf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->eot_flag) (void)hipHostFree(c->eot_flag);
    if (c->eot_flag_dev) (void)hipFree(c->eot_flag_dev);
    if (c->pair_err) (void)hipHostFree(c->pair_err);
    if (c->range_flag) (void)hipHostFree(c->range_flag);
    delete c;
}

const char* hg_last_error(hg_ctx* c) { return c ? c->err.c_str() : "null context"; }

// Test hook: out[M,N] (fp32) (+)= epilogue(A[M,K] x W[N,K]^T) with the operands rounded to fp16 on the device.
// kernel: 0 = dispatcher's choice, 1 = simple 128x128 kernel, 2 = persistent ring kernel.
int hg_preprocess_crops(hg_ctx* c, const uint8_t* img, int H, int W, const int32_t* boxes_host, int n, int n_px,
                        int pad_square, uint32_t background, float* out, uint8_t* out_u8, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (n == 0) return HG_OK;
    if (!img || !boxes_host || !out || n < 0 || H <= 0 || W <= 0 || n_px <= 0 || n_px > 4096)
        return fail(c, HG_ERR_INVALID, "hg_preprocess_crops: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    // host part: geometry of every crop (sizes, padding, resize target, centre-crop offsets, tap counts, the
    // source rows the vertical pass needs); the weight tables themselves are filled on the device
    std::vector<int32_t> head((size_t)n * (HG_PRE_HDR + 1), 0);      // headers, then the n table offsets
    int32_t* off = head.data() + (size_t)n * HG_PRE_HDR;
    size_t words = 0, tmp = 0;
    int max_rows = 0;
    for (int b = 0; b < n; ++b) {
        const int32_t* bx = boxes_host + 4 * (size_t)b;
        const int cw = bx[2] - bx[0], ch = bx[3] - bx[1];
        if (cw <= 0 || ch <= 0) return fail(c, HG_ERR_INVALID, "hg_preprocess_crops: empty box %d", b);
        int sw = cw, sh = ch, px = 0, py = 0;
        if ((pad_square & HG_PRE_PAD_SQUARE) && cw != ch) {                       // expand2square: centred, floor((side - short) / 2)
            if (cw > ch) py = (cw - ch) / 2; else px = (ch - cw) / 2;
            sw = sh = cw > ch ? cw : ch;
        }
        // torchvision Resize: short side -> n_px, long side -> int(n_px * long / short); CenterCrop offsets
        // int(round(d / 2.0)) with Python's round-half-to-even
        int nw, nh;
        if (pad_square & HG_PRE_STRETCH) { nw = nh = n_px; }        // IResize([n_px, n_px]): both sides, no centre crop
        else if (sw <= sh) { nw = n_px; nh = (int)((double)((long long)n_px * sh) / (double)sw); }
        else { nw = (int)((double)((long long)n_px * sw) / (double)sh); nh = n_px; }
        const int left = (int)nearbyint((double)(nw - n_px) / 2.0), top = (int)nearbyint((double)(nh - n_px) / 2.0);
        auto taps = [](int in, int outn) {
            const double sc = (double)in / (double)outn, fs = sc < 1.0 ? 1.0 : sc;
            return (int)ceil(2.0 * fs) * 2 + 1;
        };
        auto first_tap = [](int in, int outn, int idx) {
            const double sc = (double)in / (double)outn, fs = sc < 1.0 ? 1.0 : sc;
            const int v = (int)(((double)idx + 0.5) * sc - 2.0 * fs + 0.5);
            return v < 0 ? 0 : v;
        };
        auto end_tap = [](int in, int outn, int idx) {
            const double sc = (double)in / (double)outn, fs = sc < 1.0 ? 1.0 : sc;
            const int v = (int)(((double)idx + 0.5) * sc + 2.0 * fs + 0.5);
            return v > in ? in : v;
        };
        const int ks_h = taps(sw, nw), ks_v = taps(sh, nh);
        const int row_lo = first_tap(sh, nh, top), n_rows = end_tap(sh, nh, top + n_px - 1) - row_lo;
        if (tmp + (size_t)n_rows * n_px * 3 >= ((size_t)1 << 31))
            return fail(c, HG_ERR_INVALID, "hg_preprocess_crops: scratch of one call >= 2 GiB; split the boxes");
        int32_t* h = head.data() + (size_t)b * HG_PRE_HDR;
        h[0] = bx[0]; h[1] = bx[1]; h[2] = cw; h[3] = ch; h[4] = px; h[5] = py; h[6] = sw; h[7] = sh;
        h[8] = ks_h; h[9] = ks_v; h[10] = row_lo; h[11] = n_rows; h[12] = (int32_t)tmp; h[13] = (int32_t)background;
        h[14] = nw; h[15] = nh; h[16] = left; h[17] = top;
        off[b] = (int32_t)words;
        words += (size_t)n_px * (4 + ks_h + ks_v);
        tmp += ((size_t)n_rows * n_px * 3 + 15) / 16 * 16;
        if (n_rows > max_rows) max_rows = n_rows;
        if (words >= ((size_t)1 << 30)) return fail(c, HG_ERR_INVALID, "hg_preprocess_crops: too many boxes in one call");
    }
    int rc = ensure(c, c->pre, tmp ? tmp : 16);
    if (!rc) rc = ensure(c, c->pretab, (head.size() + words) * 4);
    if (rc) return rc;
    int32_t* head_d = (int32_t*)c->pretab.p;                          // [n][HG_PRE_HDR] | off[n] | tables
    int32_t* tab_off = head_d + (size_t)n * HG_PRE_HDR;
    int32_t* tab = tab_off + n;
    HG_HIP(hipMemcpyAsync(head_d, head.data(), head.size() * 4, hipMemcpyHostToDevice, s));
    HG_HIP(hipStreamSynchronize(s));      // `head` is a stack-lifetime host buffer
    HG_HIP(launch_preprocess(img, H, W, head_d, tab, tab_off, n, n_px, max_rows, (uint8_t*)c->pre.p, out, out_u8, s,
                             (pad_square & HG_PRE_IMAGENET_NORM) != 0));
    return HG_OK;
}

int hg_test_gemm(hg_ctx* c, const float* a, const float* w, const float* bias, float* out, int M, int N, int K,
                 int epi, int kernel, void* stream) {
    if (!c || !a || !w || !out || M <= 0) return HG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    int rc = ensure(c, c->h, rup(M, 256) * K * 2);
    if (!rc) rc = ensure(c, c->att, (size_t)N * K * 2);
    const bool f16out = (epi == EPI_BIAS_F16 || epi == EPI_BIAS_QGELU_F16 || epi == EPI_BIAS_RELU_F16);
    if (!rc && f16out) rc = ensure(c, c->qkv, rup(M, 256) * N * 2);
    // EPI_RESID_LN_F32 (timing only): centred fp16 copy into the qkv buffer, partial statistics + zero centres into cq
    const bool rln = (epi == EPI_RESID_LN_F32);
    const int sld = 4 * N / 256;
    if (!rc && rln) rc = ensure(c, c->qkv, rup(M, 256) * N * 2);
    if (!rc && rln) rc = ensure(c, c->cq, rup(M, 256) * (size_t)(2 * sld + 1) * 4);
    if (rc) return rc;
    HG_HIP(launch_f32_to_f16(a, (half_t*)c->h.p, (size_t)M * K, s));
    HG_HIP(launch_f32_to_f16(w, (half_t*)c->att.p, (size_t)N * K, s));
    GemmArgs g{};
    g.A = (half_t*)c->h.p; g.lda = K; g.W = (half_t*)c->att.p; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.out = f16out ? c->qkv.p : (void*)out; g.ldc = N;
    if (rln) {
        g.out2 = (half_t*)c->qkv.p; g.stats = (float*)c->cq.p; g.stats_ld = sld;
        g.mu = (float*)c->cq.p + (size_t)rup(M, 256) * 2 * sld;
        HG_HIP(hipMemsetAsync((void*)g.mu, 0, (size_t)M * 4, s));
    }
    hipError_t e;
    ProfScope ps(c, s, epi, M, N, K);
    if (kernel == 1) e = launch_gemm_simple(epi, g, s);
    else if (kernel == 2) e = gemm_ring_ok(g) ? launch_gemm_ring(epi, g, s) : hipErrorInvalidValue;
    else if (kernel == 3) e = gemm_duo_ok(epi, g) ? launch_gemm_duo(epi, g, s) : hipErrorInvalidValue;
    else e = launch_gemm(epi, g, s);
    ps.finish();
    if (e != hipSuccess) return fail(c, HG_ERR_HIP, "test gemm launch failed: %s", hipGetErrorString(e));
    if (f16out) HG_HIP(launch_f16_to_f32((const half_t*)c->qkv.p, out, (size_t)M * N, s));
    return HG_OK;
}

int hg_test_gemm_ln(hg_ctx* c, const float* a, const float* w, const float* bias, float* out, int M, int N, int K, int epi,
                    int kernel, const float* cs, const float* mr, const float* mu, const float* scale, float* out2,
                    float* mr_out, float* mu_out, void* stream) {
    if (!c || !a || !w || !out || M <= 0) return HG_ERR_INVALID;
    const bool lnc = (epi == EPI_LN_BIAS_F16 || epi == EPI_LN_BIAS_QGELU_F16);
    const bool rln = (epi == EPI_RESID_LN_F32 || epi == EPI_SCALE_RESID_LN_F32);
    if (!lnc && !rln) return fail(c, HG_ERR_INVALID, "hg_test_gemm_ln: epi must be 8, 9, 10 or 12");
    if (lnc && (!cs || !mr)) return fail(c, HG_ERR_INVALID, "hg_test_gemm_ln: epi 8/9 need cs and mr");
    if (rln && (!mu || !out2 || !mr_out || !mu_out || (epi == EPI_SCALE_RESID_LN_F32 && !scale)))
        return fail(c, HG_ERR_INVALID, "hg_test_gemm_ln: epi 10/12 need mu, out2, mr_out, mu_out (12: scale)");
    if (N % 256) return fail(c, HG_ERR_INVALID, "hg_test_gemm_ln: N must be a multiple of 256");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const size_t Mp = rup(M, 256);
    const int sld = 4 * (N / 256);
    int rc = ensure(c, c->h, Mp * K * 2);
    if (!rc) rc = ensure(c, c->att, (size_t)N * K * 2);
    if (!rc) rc = ensure(c, c->qkv, Mp * N * 2);
    if (!rc) rc = ensure(c, c->mr, Mp * 2 * 4);
    if (!rc) rc = ensure(c, c->mu, Mp * 4);
    if (!rc) rc = ensure(c, c->stats, Mp * (size_t)sld * 2 * 4);
    if (rc) return rc;
    HG_HIP(launch_f32_to_f16(a, (half_t*)c->h.p, (size_t)M * K, s));
    HG_HIP(launch_f32_to_f16(w, (half_t*)c->att.p, (size_t)N * K, s));
    GemmArgs g{};
    g.A = (half_t*)c->h.p; g.lda = K; g.W = (half_t*)c->att.p; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = N;
    if (lnc) {
        HG_HIP(hipMemsetAsync(c->mr.p, 0, Mp * 2 * 4, s));                  // padded rows are read by the tile's DMA
        HG_HIP(hipMemcpyAsync(c->mr.p, mr, (size_t)M * 2 * 4, hipMemcpyDeviceToDevice, s));
        g.cs = cs; g.mr = (const float*)c->mr.p; g.out = c->qkv.p;
    } else {
        HG_HIP(hipMemcpyAsync(c->mu.p, mu, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
        g.out = out; g.out2 = (half_t*)c->qkv.p; g.stats = (float*)c->stats.p; g.stats_ld = sld; g.mu = (const float*)c->mu.p;
        g.pos = scale;
    }
    hipError_t e;
    if (kernel == 2) {
        ProfScope ps(c, s, epi, M, N, K);
        e = gemm_ln_ok(epi, g) ? launch_gemm_ring(epi, g, s) : hipErrorInvalidValue;
    } else if (kernel == 3) e = gemm_duo_ok(epi, g) ? launch_gemm_duo(epi, g, s) : hipErrorInvalidValue;
    else e = launch_gemm(epi, g, s);
    if (e != hipSuccess) return fail(c, HG_ERR_HIP, "test gemm (ln) launch failed: %s", hipGetErrorString(e));
    if (lnc) {
        HG_HIP(launch_f16_to_f32((const half_t*)c->qkv.p, out, (size_t)M * N, s));
    } else {
        HG_HIP(launch_f16_to_f32((const half_t*)c->qkv.p, out2, (size_t)M * N, s));
        HG_HIP(launch_finalize_stats((const float*)c->stats.p, (float*)c->mr.p, (float*)c->mu.p, M, sld, 64, s));
        HG_HIP(hipMemcpyAsync(mr_out, c->mr.p, (size_t)M * 2 * 4, hipMemcpyDeviceToDevice, s));
        HG_HIP(hipMemcpyAsync(mu_out, c->mu.p, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
    }
    return HG_OK;
}

int hg_test_gemm_hilo(hg_ctx* c, const float* a, const float* w, const float* bias, float* x, int M, int N, int K, int steps,
                      int hilo, float* mu, float* out2, float* mr_out, void* stream) {
    if (!c || !a || !w || !x || !mu || M <= 0 || steps < 1 || steps > 16) return HG_ERR_INVALID;
    if (N % 256) return fail(c, HG_ERR_INVALID, "hg_test_gemm_hilo: N must be a multiple of 256");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const size_t Mp = rup(M, 256);
    const int sld = 4 * (N / 256);
    int rc = ensure(c, c->h, Mp * K * 2);
    if (!rc) rc = ensure(c, c->att, (size_t)N * K * 2);
    if (!rc) rc = ensure(c, c->qkv, Mp * N * 2);
    if (!rc) rc = ensure(c, c->mr, Mp * 2 * 4);
    if (!rc) rc = ensure(c, c->mu, Mp * 4);
    if (!rc) rc = ensure(c, c->muc, Mp * 4);
    if (!rc) rc = ensure(c, c->stats, Mp * (size_t)sld * 2 * 4);
    if (!rc) rc = ensure(c, c->xlo, gemm_lo_bytes(M, N));
    if (rc) return rc;
    HG_HIP(launch_f32_to_f16(a, (half_t*)c->h.p, (size_t)M * K, s));
    HG_HIP(launch_f32_to_f16(w, (half_t*)c->att.p, (size_t)N * K, s));
    HG_HIP(hipMemcpyAsync(c->mu.p, mu, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
    GemmArgs g{};
    g.A = (half_t*)c->h.p; g.lda = K; g.W = (half_t*)c->att.p; g.bias = bias; g.M = M; g.N = N; g.K = K; g.ldc = N;
    g.out = x; g.out2 = (half_t*)c->qkv.p; g.stats = (float*)c->stats.p; g.stats_ld = sld; g.mu = (const float*)c->mu.p;
    g.lo = (half_t*)c->xlo.p; g.muc = (const float*)c->muc.p;
    if (!gemm_ring2_ok(g)) return fail(c, HG_ERR_INVALID, "hg_test_gemm_hilo: shape not eligible for gemm_ring2");
    for (int i = 0; i < steps; ++i) {
        g.hl = (hilo && steps >= 2) ? (i == 0 ? 1 : (i == steps - 1 ? 3 : 2)) : 0;
        ProfScope ps(c, s, EPI_RESID_LN_F32, M, N, K);
        hipError_t e = launch_gemm_ring2(EPI_RESID_LN_F32, g, s);
        ps.finish();
        if (e != hipSuccess) return fail(c, HG_ERR_HIP, "test gemm (hi / lo) launch failed: %s", hipGetErrorString(e));
        HG_HIP(launch_finalize_stats((const float*)c->stats.p, (float*)c->mr.p, (float*)c->mu.p, M, sld, 64, s, (float*)c->muc.p));
    }
    if (out2) HG_HIP(launch_f16_to_f32((const half_t*)c->qkv.p, out2, (size_t)M * N, s));
    if (mr_out) HG_HIP(hipMemcpyAsync(mr_out, c->mr.p, (size_t)M * 2 * 4, hipMemcpyDeviceToDevice, s));
    HG_HIP(hipMemcpyAsync(mu, c->mu.p, (size_t)M * 4, hipMemcpyDeviceToDevice, s));
    return HG_OK;
}

int hg_test_attention(hg_ctx* c, const float* qkv, const float* q0, const int32_t* sel, int n_seq, int L, int heads,
                      int causal, float* out, void* stream) {
    if (!c || !qkv || !out || n_seq <= 0 || L < 1 || L > 224 || heads < 1) return HG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const int D = heads * 64;
    const size_t M = (size_t)n_seq * L;
    int rc = ensure(c, c->qkv, rup(M, 256) * 3 * D * 2);
    if (!rc) rc = ensure(c, c->att, rup(M, 256) * D * 2);
    if (!rc && q0) rc = ensure(c, c->cq, rup(n_seq, 256) * (size_t)D * 2);
    if (rc) return rc;
    HG_HIP(launch_f32_to_f16(qkv, (half_t*)c->qkv.p, M * 3 * D, s));
    if (q0) {      // one query row per sequence (row sel[seq], or 0): out [n_seq, D]
        HG_HIP(launch_f32_to_f16(q0, (half_t*)c->cq.p, (size_t)n_seq * D, s));
        HG_HIP(launch_attention_row0((const half_t*)c->qkv.p, (const half_t*)c->cq.p, sel, (half_t*)c->att.p, n_seq, L,
                                     heads, (causal & 1) != 0, s));
        HG_HIP(launch_f16_to_f32((const half_t*)c->att.p, out, (size_t)n_seq * D, s));
    } else {
        // (causal bit 1: the one-workgroup-per-item launch for L <= 32 instead of four items per workgroup - same bits: tests)
        HG_HIP(launch_attention((const half_t*)c->qkv.p, (half_t*)c->att.p, n_seq, L, heads, (causal & 1) != 0, s, 0, !(causal & 2)));
        HG_HIP(launch_f16_to_f32((const half_t*)c->att.p, out, M * D, s));
    }
    return HG_OK;
}

int hg_test_qkv_attn(hg_ctx* c, const float* a, const float* w, const float* bias, const float* cs, const float* mr, int n_seq,
                     int L, int heads, int fused, float* out, void* stream) {
    if (!c || !a || !w || !cs || !mr || !out || n_seq <= 0 || heads < 1) return HG_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const int D = heads * 64, M = n_seq * L;
    const int K = D + ((fused & 2) ? 64 : 0);      // (bit 1: a is [M, D + 64], w [3D, D + 64] - the shape of a block with a folded adapter)
    fused &= 1;
    const size_t Mp = rup(M, 256);
    if (fused && !qkv_attn_ok(n_seq, L, D, heads, K, K)) return fail(c, HG_ERR_INVALID, "hg_test_qkv_attn: shape not eligible for the fused kernel");
    int rc = ensure(c, c->h, Mp * K * 2);
    if (!rc) rc = ensure(c, c->fc, (size_t)3 * D * K * 2 * 2 + (size_t)(heads / 2 + 1) * 768 * 4);
    if (!rc) rc = ensure(c, c->qkv, Mp * 3 * D * 2);
    if (!rc) rc = ensure(c, c->att, Mp * D * 2);
    if (!rc) rc = ensure(c, c->mr, Mp * 2 * 4);
    if (rc) return rc;
    half_t* w16 = (half_t*)c->fc.p;
    half_t* wp = w16 + (size_t)3 * D * K;
    float* bcs = (float*)(wp + (size_t)3 * D * K);
    HG_HIP(hipMemsetAsync(c->h.p, 0, Mp * K * 2, s));
    HG_HIP(launch_f32_to_f16(a, (half_t*)c->h.p, (size_t)M * K, s));
    HG_HIP(launch_f32_to_f16(w, w16, (size_t)3 * D * K, s));
    HG_HIP(hipMemsetAsync(c->mr.p, 0, Mp * 2 * 4, s));
    HG_HIP(hipMemcpyAsync(c->mr.p, mr, (size_t)M * 2 * 4, hipMemcpyDeviceToDevice, s));
    HG_HIP(hipMemsetAsync(c->att.p, 0, Mp * D * 2, s));
    if (fused) {
        HG_HIP(launch_pack_qkv(w16, bias, cs, wp, bcs, D, heads, s, K));
        QkvAttnArgs qa{};
        qa.x16 = (const half_t*)c->h.p; qa.lda = K; qa.K = K; qa.wp = wp; qa.bcs = bcs; qa.mr = (const float*)c->mr.p;
        qa.out = (half_t*)c->att.p; qa.ldo = D; qa.n_seq = n_seq; qa.L = L; qa.D = D; qa.heads = heads;
        qa.gsz = c->opt_qkv_attn_gsz; qa.a_bytes = (unsigned)(Mp * (size_t)K * 2);
#ifdef HG_STAMPS
        if (!(rc = ensure(c, c->cq, 256 * 8 * 16 * 8))) qa.dbg = (unsigned long long*)c->cq.p;      // read back by tools/qkv_attn_stamps.py
        else return rc;
        HG_HIP(hipMemsetAsync(c->cq.p, 0, 256 * 8 * 16 * 8, s));
#endif
        ProfScope ps(c, s, HG_PROF_QKV_ATTN, n_seq, L, heads);
        hipError_t e = launch_qkv_attn(qa, s);
        ps.finish();
        if (e != hipSuccess) return fail(c, HG_ERR_HIP, "test qkv_attn launch failed: %s", hipGetErrorString(e));
#ifdef HG_STAMPS
        {      // diagnostic build: median over workgroups of the per-wave s_memtime totals of every phase (hg_qkv_attn.hip QA_ST)
            HG_HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> hd((size_t)256 * 8 * 16);
            HG_HIP(hipMemcpy(hd.data(), c->cq.p, hd.size() * 8, hipMemcpyDeviceToHost));
            static const char* nm[14] = {"K loop", "drain+barrier", "barrier behind head a", "attention a", "barrier", "head b -> LDS",
                                         "attention b", "barrier", "whole kernel", "LN fold", "head a -> LDS", "K loop: wait A", "K loop: barrier",
                                         "K loop: wait W"};
            for (int wv : {0, 3, 4, 6, 7}) {
                fprintf(stderr, "[stamps] wave %d:", wv);
                for (int k = 0; k < 14; ++k) {
                    std::vector<unsigned long long> v;
                    for (int b = 0; b < 256; ++b) if (hd[((size_t)b * 8 + wv) * 16 + 8]) v.push_back(hd[((size_t)b * 8 + wv) * 16 + k]);
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, " %s %llu |", nm[k], v[v.size() / 2]);
                }
                fprintf(stderr, "\n");
            }
        }
#endif
    } else {
        GemmArgs g{};
        g.A = (const half_t*)c->h.p; g.lda = K; g.W = w16; g.bias = bias; g.cs = cs; g.mr = (const float*)c->mr.p;
        g.out = c->qkv.p; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = K;
        if (!gemm_ln_ok(EPI_LN_BIAS_F16, g)) return fail(c, HG_ERR_INVALID, "hg_test_qkv_attn: shape not eligible for the folded GEMM");
        HG_HIP(gemm(c, EPI_LN_BIAS_F16, g, s));
        HG_HIP(attention(c, (const half_t*)c->qkv.p, (half_t*)c->att.p, n_seq, L, heads, false, s));
    }
    HG_HIP(launch_f16_to_f32((const half_t*)c->att.p, out, (size_t)M * D, s));
    return HG_OK;
}

int hg_profile_begin(hg_ctx* c, int kind, int max_launches) {
    if (!c || max_launches < 0) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    c->prof_ev.clear();
    c->prof_rec.clear();
    c->prof_n = 0;
    c->prof_kind = kind;
    if (kind == HG_PROF_OFF) return HG_OK;
    c->prof_ev.resize((size_t)2 * max_launches);
    c->prof_rec.resize((size_t)max_launches);
    for (auto& e : c->prof_ev) HG_HIP(hipEventCreate(&e));
    return HG_OK;
}

int hg_profile_end(hg_ctx* c, hg_prof_rec* recs, int max_recs, int32_t* n_recs) {
    if (!c || !n_recs || (max_recs > 0 && !recs)) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    int n = 0;
    for (size_t i = 0; i < c->prof_n && n < max_recs; ++i, ++n) {
        HG_HIP(hipEventSynchronize(c->prof_ev[2 * i + 1]));
        float ms = 0.f;
        HG_HIP(hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        recs[n] = c->prof_rec[i];
        recs[n].ms = ms;
    }
    *n_recs = n;
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    c->prof_ev.clear();
    c->prof_rec.clear();
    c->prof_kind = HG_PROF_OFF;
    c->prof_n = 0;
    return HG_OK;
}

int hg_workspace_bytes(hg_ctx* c, uint64_t* bytes) {
    if (!c || !bytes) return HG_ERR_INVALID;
    Buf* bufs[] = {&c->x, &c->h, &c->qkv, &c->att, &c->fc, &c->head16, &c->tok32, &c->small, &c->i32,
                   &c->ad32, &c->ad16, &c->adkv, &c->mr, &c->mu, &c->muc, &c->stats, &c->pre, &c->pretab, &c->cx, &c->ca, &c->ch, &c->cf, &c->cq, &c->xlo, &c->zpark, &c->att2, &c->hg, &c->pair_ready};
    uint64_t t = 0;
    for (Buf* b : bufs) t += b->bytes;
    *bytes = t;
    return HG_OK;
}

// ---- weights --------------------------------------------------------------------------------------------
int hg_load_vit(hg_ctx* c, const hg_vit_weights* w) {
    if (!c || !w) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    Vit& v = c->vit;
    free_all(v.owned);
    free_all(v.owned_adapters);
    v = Vit{};
    const int D = w->width, p = w->patch_size;
    if (D <= 0 || D % 128 || w->heads * 64 != D)
        return fail(c, HG_ERR_INVALID, "vision width must be a multiple of 128 with heads = width/64 (got %d, %d)", D,
                    w->heads);
    if (p <= 0 || p % 8 || w->input_resolution % p || (3 * p * p) % 64)
        return fail(c, HG_ERR_INVALID, "unsupported patch size %d / resolution %d", p, w->input_resolution);
    if (w->output_dim <= 0 || w->output_dim % 128)
        return fail(c, HG_ERR_INVALID, "output_dim must be a multiple of 128 (got %d)", w->output_dim);
    v.D = D; v.layers = w->layers; v.heads = w->heads; v.patch = p; v.res = w->input_resolution;
    v.grid = v.res / p; v.L = v.grid * v.grid + 1; v.E = w->output_dim; v.Kp = 3 * p * p;
    if (v.L > 224) return fail(c, HG_ERR_INVALID, "at most 224 tokens per image supported (got %d)", v.L);
    int rc = 0;
    keep_first(rc, as_f16(c, v.owned, w->conv1_weight, (size_t)D * v.Kp, &v.w_patch, "visual.conv1.weight"));
    keep_first(rc, as_f32(c, v.owned, w->class_embedding, D, &v.cls, "visual.class_embedding"));
    keep_first(rc, as_f32(c, v.owned, w->positional_embedding, (size_t)v.L * D, &v.pos, "visual.positional_embedding"));
    keep_first(rc, as_f32(c, v.owned, w->ln_pre_weight, D, &v.lnpre_w, "visual.ln_pre.weight"));
    keep_first(rc, as_f32(c, v.owned, w->ln_pre_bias, D, &v.lnpre_b, "visual.ln_pre.bias"));
    keep_first(rc, as_f32(c, v.owned, w->ln_post_weight, D, &v.lnpost_w, "visual.ln_post.weight"));
    keep_first(rc, as_f32(c, v.owned, w->ln_post_bias, D, &v.lnpost_b, "visual.ln_post.bias"));
    keep_first(rc, as_f16_T(c, v.owned, w->proj, D, v.E, &v.w_projT, "visual.proj"));
    if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
    rc = load_blocks(c, v.owned, w->blocks, v.layers, D, v.blocks, true);
    if (rc) return rc;
    rc = load_adapters(c, w->adapters, v.layers);
    if (rc) return rc;
    HG_HIP(hipDeviceSynchronize());
    v.loaded = true;
    return HG_OK;
}

int hg_update_adapters(hg_ctx* c, const hg_adapter_weights* adapters, int layers) {
    if (!c) return HG_ERR_INVALID;
    if (!c->vit.loaded) return fail(c, HG_ERR_NOT_LOADED, "hg_load_vit first");
    HG_ON_DEVICE(c);
    HG_HIP(hipDeviceSynchronize());
    return load_adapters(c, adapters, layers);
}

int hg_load_text(hg_ctx* c, const hg_text_weights* w) {
    if (!c || !w) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    Text& t = c->text;
    free_all(t.owned);
    t = Text{};
    const int D = w->width;
    if (D <= 0 || D % 128 || w->heads * 64 != D)
        return fail(c, HG_ERR_INVALID, "text width must be a multiple of 128 with heads = width/64 (got %d, %d)", D,
                    w->heads);
    if (w->context_length > 224) return fail(c, HG_ERR_INVALID, "context_length > 224 unsupported");
    if (w->output_dim <= 0 || w->output_dim % 128) return fail(c, HG_ERR_INVALID, "output_dim %% 128 != 0");
    t.D = D; t.layers = w->layers; t.heads = w->heads; t.ctx = w->context_length; t.vocab = w->vocab_size;
    t.E = w->output_dim;
    int rc = 0;
    keep_first(rc, as_f32(c, t.owned, w->token_embedding, (size_t)t.vocab * D, &t.tok, "token_embedding.weight"));
    keep_first(rc, as_f32(c, t.owned, w->positional_embedding, (size_t)t.ctx * D, &t.pos, "positional_embedding"));
    keep_first(rc, as_f32(c, t.owned, w->ln_final_weight, D, &t.lnf_w, "ln_final.weight"));
    keep_first(rc, as_f32(c, t.owned, w->ln_final_bias, D, &t.lnf_b, "ln_final.bias"));
    keep_first(rc, as_f16_T(c, t.owned, w->text_projection, D, t.E, &t.w_projT, "text_projection"));
    if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
    rc = load_blocks(c, t.owned, w->blocks, t.layers, D, t.blocks, true);      // (folded operands too: option text_ln_fold)
    if (rc) return rc;
    HG_HIP(hipDeviceSynchronize());
    t.loaded = true;
    return HG_OK;
}

int hg_load_vae(hg_ctx* c, int slot, const hg_vae_weights* w) {
    if (!c || !w || slot < 0 || slot >= HG_MAX_SLOTS) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    Vae& v = c->vae[slot];
    free_all(v.owned);
    v = Vae{};
    v.dim = w->dim; v.eh = w->enc_hidden; v.gh = w->gen_hidden;
    if (v.dim <= 0 || v.dim % 128) return fail(c, HG_ERR_INVALID, "vae dim must be a multiple of 128");
    int rc = 0;
    if (w->enc_w0.ptr) {
        if (v.eh <= 0 || v.eh % 128) return fail(c, HG_ERR_INVALID, "enc_hidden must be a multiple of 128");
        keep_first(rc, as_f16(c, v.owned, w->enc_w0, (size_t)v.eh * v.dim, &v.e_w0, "Encoder.net.0.weight"));
        keep_first(rc, as_f32(c, v.owned, w->enc_b0, v.eh, &v.e_b0, "Encoder.net.0.bias"));
        // mean | log_var stacked into one [2*dim, eh] GEMM operand
        void* p;
        keep_first(rc, dev_alloc(c, v.owned, (size_t)2 * v.dim * v.eh * 2, &p));
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        v.e_wml = (half_t*)p;
        std::vector<void*> sc;
        half_t *m, *l;
        keep_first(rc, as_f16(c, sc, w->enc_mean_w, (size_t)v.dim * v.eh, &m, "Encoder.mean.weight"));
        keep_first(rc, as_f16(c, sc, w->enc_logvar_w, (size_t)v.dim * v.eh, &l, "Encoder.log_var.weight"));
        if (!rc) {
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(v.e_wml, m, (size_t)v.dim * v.eh * 2, hipMemcpyDeviceToDevice);
            (void)hipMemcpy(v.e_wml + (size_t)v.dim * v.eh, l, (size_t)v.dim * v.eh * 2, hipMemcpyDeviceToDevice);
        }
        free_all(sc);
        keep_first(rc, dev_alloc(c, v.owned, (size_t)2 * v.dim * 4, &p));
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        v.e_bml = (float*)p;
        float *bm, *bl;
        keep_first(rc, as_f32(c, sc, w->enc_mean_b, v.dim, &bm, "Encoder.mean.bias"));
        keep_first(rc, as_f32(c, sc, w->enc_logvar_b, v.dim, &bl, "Encoder.log_var.bias"));
        if (!rc) {
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(v.e_bml, bm, (size_t)v.dim * 4, hipMemcpyDeviceToDevice);
            (void)hipMemcpy(v.e_bml + v.dim, bl, (size_t)v.dim * 4, hipMemcpyDeviceToDevice);
        }
        free_all(sc);
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        v.enc = true;
    }
    if (w->gen_w0.ptr) {
        if (v.gh <= 0 || v.gh % 128) return fail(c, HG_ERR_INVALID, "gen_hidden must be a multiple of 128");
        keep_first(rc, as_f16(c, v.owned, w->gen_w0, (size_t)v.gh * v.dim, &v.g_w0, "Generator.net.0.weight"));
        keep_first(rc, as_f32(c, v.owned, w->gen_b0, v.gh, &v.g_b0, "Generator.net.0.bias"));
        keep_first(rc, as_f16(c, v.owned, w->gen_w2, (size_t)v.dim * v.gh, &v.g_w2, "Generator.net.2.weight"));
        keep_first(rc, as_f32(c, v.owned, w->gen_b2, v.dim, &v.g_b2, "Generator.net.2.bias"));
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        v.gen = true;
    }
    // the one-kernel path's operand: the same fp16 weights as a linear stream of MFMA fragments in order of use
    if (vae_fused_ok(v.dim, v.enc ? v.eh : 0, v.gen ? v.gh : 0)) {
        const size_t bytes = (v.enc ? 2 * vae_fused_pass_bytes(v.eh) : 0) + (v.gen ? vae_fused_pass_bytes(v.gh) : 0);
        void* p;
        rc = dev_alloc(c, v.owned, bytes, &p);
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
        v.wp = (half_t*)p;
        HG_HIP(launch_pack_vae(v.enc ? v.e_w0 : nullptr, v.e_wml, v.eh, v.gen ? v.g_w0 : nullptr, v.g_w2, v.gh, v.wp, nullptr));
    }
    HG_HIP(hipDeviceSynchronize());
    return HG_OK;
}

int hg_load_mlp(hg_ctx* c, int slot, const hg_mlp_weights* w) {
    if (!c || !w || slot < 0 || slot >= HG_MAX_SLOTS) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    Mlp& m = c->mlp[slot];
    free_all(m.owned);
    m = Mlp{};
    m.in = w->in_dim; m.hid = w->hidden_dim; m.out = w->out_dim;
    if (m.in % 64 || m.hid % 128 || m.out % 128 || m.in <= 0) return fail(c, HG_ERR_INVALID, "mlp_net dims must be multiples of 128");
    int rc = 0;
    keep_first(rc, as_f16(c, m.owned, w->w0, (size_t)m.hid * m.in, &m.w0, "mlp.net.0.weight"));
    keep_first(rc, as_f32(c, m.owned, w->b0, m.hid, &m.b0, "mlp.net.0.bias"));
    keep_first(rc, as_f16(c, m.owned, w->w2, (size_t)m.hid * m.hid, &m.w2, "mlp.net.2.weight"));
    keep_first(rc, as_f32(c, m.owned, w->b2, m.hid, &m.b2, "mlp.net.2.bias"));
    keep_first(rc, as_f16(c, m.owned, w->w4, (size_t)m.out * m.hid, &m.w4, "mlp.net.4.weight"));
    keep_first(rc, as_f32(c, m.owned, w->b4, m.out, &m.b4, "mlp.net.4.bias"));
    if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
    HG_HIP(hipDeviceSynchronize());
    m.loaded = true;
    return HG_OK;
}

int hg_roi_align(hg_ctx* c, const float* feat, int C, int H, int W, const float* boxes, int n, float spatial_scale,
                 int P, float* out_pooled, float* out_mean, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (n == 0) return HG_OK;
    if (!feat || !boxes || n < 0 || C <= 0 || H <= 0 || W <= 0 || P <= 0 || (!out_pooled && !out_mean))
        return fail(c, HG_ERR_INVALID, "hg_roi_align: bad arguments");
    HG_ON_DEVICE(c);
    HG_HIP(launch_roi_align(feat, C, H, W, boxes, n, spatial_scale, P, out_pooled, out_mean, (hipStream_t)stream));
    return HG_OK;
}

// ---- cache-model logits -----------------------------------------------------------------
static int to_host_f32(hg_ctx* c, const hg_tensor& t, size_t n, std::vector<float>& out, const char* name) {
    std::vector<void*> sc;
    float* d = nullptr;
    int rc = as_f32(c, sc, t, n, &d, name);
    if (rc) { free_all(sc); return rc; }
    out.resize(n);
    hipError_t e = hipMemcpy(out.data(), d, n * 4, hipMemcpyDeviceToHost);
    free_all(sc);
    return e == hipSuccess ? HG_OK : fail(c, HG_ERR_HIP, "hipMemcpy D2H failed for %s", name);
}

namespace {
int upload_f32(hg_ctx* c, std::vector<void*>& owned, const std::vector<float>& v, float** out) {
    void* p;
    int rc = dev_alloc(c, owned, v.size() * 4, &p);
    if (rc) return rc;
    HG_HIP(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return HG_OK;
}
}  // namespace

int hg_load_cache(hg_ctx* c, int slot, const hg_cache_weights* w) {
    if (!c || !w || slot < 0 || slot >= HG_MAX_CACHE_SLOTS) return HG_ERR_INVALID;
    HG_ON_DEVICE(c);
    Cache& m = c->cache[slot];
    free_all(m.owned);
    m = Cache{};
    m.S = w->S; m.K = w->K; m.C = w->C; m.has_labels = w->labels.ptr != nullptr;
    if (m.S <= 0 || m.K <= 0 || m.K % 64 || (m.has_labels && m.C <= 0))
        return fail(c, HG_ERR_INVALID, "cache model: S > 0, K %% 64 == 0 (got S=%d K=%d C=%d)", m.S, m.K, m.C);
    m.Sp = (int)rup(m.S, 128);
    m.Cp = m.has_labels ? (int)rup(m.C, 128) : 0;
    // weight rows padded with zeros to a multiple of 128 (the GEMM's N granularity)
    int rc = dev_alloc(c, m.owned, (size_t)m.Sp * m.K * 2, (void**)&m.w16);
    if (rc) return rc;
    HG_HIP(hipMemset(m.w16, 0, (size_t)m.Sp * m.K * 2));
    if (!w->weight.ptr) return fail(c, HG_ERR_INVALID, "missing tensor cache weight");
    if (w->weight.dtype == HG_F16) HG_HIP(hipMemcpy(m.w16, w->weight.ptr, (size_t)m.S * m.K * 2, hipMemcpyDeviceToDevice));
    else HG_HIP(launch_f32_to_f16((const float*)w->weight.ptr, m.w16, (size_t)m.S * m.K, 0));
    std::vector<float> bias(m.Sp, 0.f);
    if (w->bias.ptr) {
        std::vector<float> b;
        rc = to_host_f32(c, w->bias, m.S, b, "cache bias");
        if (rc) return rc;
        for (int i = 0; i < m.S; ++i) bias[i] = b[i];
    }
    if (!m.has_labels) {
        rc = upload_f32(c, m.owned, bias, &m.b);
        if (rc) return rc;
    } else {
        // (f W^T + b) L / lens / post_div = ((f W^T) L + b L) * scale: the bias term is a per-class constant
        // (kept in fp32; phi = f W^T alone goes through fp16 for the second MFMA GEMM)
        std::vector<float> lab, lens;
        rc = to_host_f32(c, w->labels, (size_t)m.S * m.C, lab, "cache labels");
        if (!rc) rc = to_host_f32(c, w->sample_lens, m.C, lens, "cache sample_lens");
        if (rc) return rc;
        std::vector<float> lt((size_t)m.Cp * m.Sp, 0.f), bc(m.Cp, 0.f), sc(m.Cp, 0.f);
        for (int cc = 0; cc < m.C; ++cc) {
            double acc = 0.0;
            for (int i = 0; i < m.S; ++i) {
                const float v = lab[(size_t)i * m.C + cc];
                lt[(size_t)cc * m.Sp + i] = v;
                acc += (double)bias[i] * v;
            }
            bc[cc] = (float)acc;
            sc[cc] = 1.0f / (lens[cc] * (w->post_div != 0.f ? w->post_div : 1.f));
        }
        float* lt32 = nullptr;
        std::vector<void*> scratch;
        rc = upload_f32(c, scratch, lt, &lt32);
        if (!rc) rc = dev_alloc(c, m.owned, lt.size() * 2, (void**)&m.lt16);
        if (!rc) { hipError_t e = launch_f32_to_f16(lt32, m.lt16, lt.size(), 0); if (e != hipSuccess) rc = HG_ERR_HIP; }
        if (!rc) { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) rc = HG_ERR_HIP; }
        free_all(scratch);
        if (!rc) rc = upload_f32(c, m.owned, bc, &m.bias_c);
        if (!rc) rc = upload_f32(c, m.owned, sc, &m.scale);
        if (rc) return rc < 0 ? rc : HG_ERR_INVALID;
    }
    HG_HIP(hipDeviceSynchronize());
    m.loaded = true;
    return HG_OK;
}

int hg_cache_logits(hg_ctx* c, int slot, const float* feats, int R, float* out, void* stream) {
    if (!c || slot < 0 || slot >= HG_MAX_CACHE_SLOTS) return HG_ERR_INVALID;
    Cache& m = c->cache[slot];
    if (!m.loaded) return fail(c, HG_ERR_NOT_LOADED, "cache slot %d not loaded", slot);
    if (R == 0) return HG_OK;
    if (R < 0 || !feats || !out) return fail(c, HG_ERR_INVALID, "bad arguments to cache_logits");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    for (int r0 = 0, Rc = 0; r0 < R; r0 += Rc) {
        Rc = chunk_rows(c, R - r0);
        const size_t Rp = rup(Rc, 256);
        const int Np = m.has_labels ? m.Cp : m.Sp, Nout = m.has_labels ? m.C : m.S;
        int rc = ensure(c, c->h, Rp * m.K * 2);
        if (!rc) rc = ensure(c, c->att, Rp * m.Sp * 2);
        if (!rc) rc = ensure(c, c->x, Rp * Np * 4);
        if (rc) return rc;
        half_t* f16 = (half_t*)c->h.p;
        half_t* phi = (half_t*)c->att.p;
        float* tmp = (float*)c->x.p;
        HG_HIP(launch_f32_to_f16(feats + (size_t)r0 * m.K, f16, (size_t)Rc * m.K, s));
        GemmArgs g{};
        g.A = f16; g.lda = m.K; g.W = m.w16; g.M = Rc; g.N = m.Sp; g.K = m.K;
        if (!m.has_labels) {
            g.bias = m.b; g.out = tmp; g.ldc = m.Sp;
            HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
        } else {
            g.bias = nullptr; g.out = phi; g.ldc = m.Sp;
            HG_HIP(gemm(c, EPI_BIAS_F16, g, s));
            HG_HIP(hipMemsetAsync(tmp, 0, (size_t)Rc * m.Cp * 4, s));
            g = GemmArgs{};
            g.A = phi; g.lda = m.Sp; g.W = m.lt16; g.bias = m.bias_c; g.pos = m.scale; g.out = tmp; g.ldc = m.Cp;
            g.M = Rc; g.N = m.Cp; g.K = m.Sp;
            HG_HIP(gemm(c, EPI_SCALE_RESID_F32, g, s));          // 0 + (phi L + b L) * scale
        }
        HG_HIP(launch_copy_cols(tmp, Np, out + (size_t)r0 * Nout, Rc, Nout, s));
    }
    return HG_OK;
}

// ---- image tower ------------------------------------------------------------------------------------------
static int encode_image_impl(hg_ctx* c, const float* x_nchw, const float* priors, const uint8_t* mask, int B, int N,
                             float* out, float* out_local, float* trace, bool variant_c, hipStream_t s) {
    if (!c) return HG_ERR_INVALID;
    Vit& v = c->vit;
    if (!v.loaded) return fail(c, HG_ERR_NOT_LOADED, "hg_load_vit has not been called");
    if (B == 0) return HG_OK;
    if (B < 0 || !x_nchw || !out) return fail(c, HG_ERR_INVALID, "bad arguments to encode_image");
    if (variant_c && !out_local) return fail(c, HG_ERR_INVALID, "out_local == NULL");
    if (priors && (N <= 0 || !mask)) return fail(c, HG_ERR_INVALID, "priors given but N <= 0 or mask == NULL");
    HG_ON_DEVICE(c);
    const int D = v.D, L = v.L, G = L - 1, E = v.E;
    const size_t img = (size_t)3 * v.res * v.res;
    for (int b0 = 0; b0 < B; b0 += c->max_chunk_img) {
        const int Bc = (B - b0 < c->max_chunk_img) ? B - b0 : c->max_chunk_img;
        const int M = Bc * L;
        int rc = ensure_tower_ws(c, M, D);
        if (!rc) rc = ensure(c, c->head16, rup(variant_c ? M : Bc, 256) * D * 2);
        if (!rc && variant_c) rc = ensure(c, c->tok32, (size_t)M * E * 4);
        if (!rc) rc = ensure(c, c->fc, rup((size_t)Bc * G, 256) * (v.Kp > 4 * D ? v.Kp : 4 * D) * 2);
        if (rc) return rc;
        float* x = (float*)c->x.p;
        half_t* patches = (half_t*)c->fc.p;
        // conv1 as GEMM over the patch matrix; epilogue scatters to token rows 1.. and adds pos
        HG_HIP(launch_im2col(x_nchw + (size_t)b0 * img, patches, Bc, v.res, v.patch, s));
        GemmArgs g{};
        g.A = patches; g.lda = v.Kp; g.W = v.w_patch; g.bias = nullptr; g.out = x; g.ldc = D;
        // class rows, positional embedding and ln_pre are one pass over the rows (run_blocks): the GEMM only scatters
        g.M = Bc * G; g.N = D; g.K = v.Kp; g.pos = nullptr; g.G = G; g.L = L;
        HG_HIP(gemm(c, EPI_PATCH_F32, g, s));
        float* tr = trace ? trace + (size_t)b0 * D : nullptr;      // (row 0 of the trace = after ln_pre: copied inside run_blocks)
        const int tstride = B * D;
        AdapterCall ac;
        ac.enabled = variant_c;
        ac.priors = priors ? priors + (size_t)b0 * N * 64 : nullptr;
        ac.mask = mask ? mask + (size_t)b0 * N : nullptr;
        ac.N = N;
        const float* row0 = nullptr;      // dense class-token rows when the last block ran on them only
        rc = run_blocks(c, v.blocks, Bc, L, D, v.heads, false, s, tr, tstride, &ac, true, variant_c ? nullptr : &row0, nullptr,
                        v.lnpre_w, v.lnpre_b, v.pos, v.cls);
        if (rc) return rc;
        half_t* h16 = (half_t*)c->head16.p;
        if (!variant_c) {
            // ln_post(x[:,0,:]) @ proj   (clipnet/model.py:231-234)
            if (row0) HG_HIP(launch_layernorm_f16(row0, v.lnpost_w, v.lnpost_b, h16, Bc, D, nullptr, 0, 1, s));
            else HG_HIP(launch_layernorm_f16(x, v.lnpost_w, v.lnpost_b, h16, Bc, D, nullptr, 0, L, s));
            g = GemmArgs{};
            g.A = h16; g.lda = D; g.W = v.w_projT; g.out = out + (size_t)b0 * E; g.ldc = E; g.M = Bc; g.N = E; g.K = D;
            HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
        } else {
            // ln_post + proj on all tokens, split into (global, local NCHW)  (adapter...:501-506)
            HG_HIP(launch_layernorm_f16(x, v.lnpost_w, v.lnpost_b, h16, M, D, nullptr, 0, 1, s));
            g = GemmArgs{};
            g.A = h16; g.lda = D; g.W = v.w_projT; g.out = c->tok32.p; g.ldc = E; g.M = M; g.N = E; g.K = D;
            HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
            HG_HIP(launch_split_global_local((const float*)c->tok32.p, out + (size_t)b0 * E,
                                             out_local + (size_t)b0 * E * G, Bc, L, E, s));
        }
    }
    return HG_OK;
}

int hg_encode_image(hg_ctx* c, const float* x_nchw, int B, float* out, void* stream) {
    return encode_image_impl(c, x_nchw, nullptr, nullptr, B, 0, out, nullptr, nullptr, false, (hipStream_t)stream);
}
int hg_encode_image_trace(hg_ctx* c, const float* x_nchw, int B, float* out, float* trace, void* stream) {
    return encode_image_impl(c, x_nchw, nullptr, nullptr, B, 0, out, nullptr, trace, false, (hipStream_t)stream);
}
int hg_encode_image_prior(hg_ctx* c, const float* x_nchw, const float* priors, const uint8_t* mask, int B, int N,
                          float* out_global, float* out_local_nchw, void* stream) {
    return encode_image_impl(c, x_nchw, priors, mask, B, priors ? N : 0, out_global, out_local_nchw, nullptr, true,
                             (hipStream_t)stream);
}

// ---- text tower -------------------------------------------------------------------------------------------
// A truncation length that does not cover every EOT position (a stale host-side max(EOT)) must not gather
// another sequence's row or read out of bounds: EOT indices are clamped into [0, Leff) on the device; the same flag turns the
// WHOLE output of that call into NaN at its end (the stale call itself is loud: launch_poison_if_flag) and, being sticky and
// host-mapped, makes the NEXT text call fail with HG_ERR_INVALID and an explanation (this call cannot be failed without a sync).
static int text_check_flag(hg_ctx* c) {
    if (c->eot_flag && *(volatile int32_t*)c->eot_flag) {
        *(volatile int32_t*)c->eot_flag = 0;
        return fail(c, HG_ERR_INVALID, "a previous encode_text call was given trunc < max(EOT)+1: its EOT rows were clamped and "
                                       "its outputs are invalid (recompute the truncation length from the current token ids)");
    }
    return HG_OK;
}

// Prompts per pass of the text tower: a budget of ROWS, so that a call truncated to the 13-16 tokens its prompts really have (the
// generation pipeline: main_tip_finetune.py:759-824) fills the GEMMs' row tiles like a full-length one does - 640 prompts of 13 tokens
// are 33 row tiles of 256 on 256 CUs.  The budget is 65 536 rows: with D = 512 every GEMM of a block then fills whole rounds (residual
// GEMMs 512 x 2 tiles of 128 x 256 on 2 x 256 workgroups; in_proj 256 x 6 and c_fc 256 x 8 tiles of 256 x 256 on 256) - the 49 280
// rows of 640 full-length prompts leave them at 1.50, 4.52 and 6.03 rounds.  A call longer than one pass is cut into EQUAL passes
// (no small tail pass).  A call that fits one pass is one pass, as before (600 prompts x 77 tokens = 46 200 rows).
static int text_chunk_prompts(const hg_ctx* c, int n_prompts, int Leff) {
    const long budget = c->text_rows_budget, L = Leff > 0 ? Leff : 1;
    long per = budget / L;
    if (per < 1) per = 1;
    if (n_prompts <= per) return n_prompts > 0 ? n_prompts : 1;
    const long passes = (n_prompts + per - 1) / per;
    return (int)((n_prompts + passes - 1) / passes);
}

static int text_tail(hg_ctx* c, int Tc, int Leff, const int32_t* eot, float* out, hipStream_t s) {
    Text& t = c->text;
    const int D = t.D, E = t.E;
    // option text_ln_fold: 1 (default) folds the LayerNorms with gamma riding in the ACTIVATION copy (6.2e-4 against the reference's
    // fixture, closer than the separate kernels' 6.5e-4, 5.35 -> 5.15 ms for 600 x 77 tokens); 2 folds gamma into the weights (4.8 ms,
    // 7.6e-4, worst prompt 9.6e-4 of the 1e-3 budget); 0 runs the separate kernels
    const float* rows = nullptr;      // dense EOT rows when the last block ran on them only
    int rc = run_blocks(c, t.blocks, Tc, Leff, D, t.heads, true, s, nullptr, 0, nullptr, c->opt_text_ln_fold != 0, &rows, eot, nullptr,
                        nullptr, nullptr, nullptr, c->opt_text_ln_fold == 1);
    if (rc) return rc;
    half_t* h16 = (half_t*)c->head16.p;
    // ln_final, select the EOT row, @ text_projection (clipnet/model.py:346-350); LN is row-wise so
    // selecting before normalising is identical
    if (rows) HG_HIP(launch_layernorm_f16(rows, t.lnf_w, t.lnf_b, h16, Tc, D, nullptr, 0, 1, s));
    else HG_HIP(launch_layernorm_f16((const float*)c->x.p, t.lnf_w, t.lnf_b, h16, Tc, D, eot, Leff, 0, s));
    GemmArgs g{};
    g.A = h16; g.lda = D; g.W = t.w_projT; g.out = out; g.ldc = E; g.M = Tc; g.N = E; g.K = D;
    HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
    return HG_OK;
}

int hg_encode_text_ids(hg_ctx* c, const int32_t* ids, int T, int L, float* out, int trunc, void* stream) {
    if (!c) return HG_ERR_INVALID;
    Text& t = c->text;
    if (!t.loaded) return fail(c, HG_ERR_NOT_LOADED, "hg_load_text has not been called");
    if (T == 0) return HG_OK;
    if (T < 0 || !ids || !out || L < 1 || L > t.ctx) return fail(c, HG_ERR_INVALID, "bad arguments to encode_text_ids");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    if (int frc = text_check_flag(c)) return frc;
    const int Leff = (trunc > 0 && trunc < L) ? trunc : L;
    if (Leff < L && c->eot_flag_dev) HG_HIP(hipMemsetAsync(c->eot_flag_dev, 0, 4, s));
    const int chunk = text_chunk_prompts(c, T, Leff);
    for (int t0 = 0; t0 < T; t0 += chunk) {
        const int Tc = (T - t0 < chunk) ? T - t0 : chunk;
        int rc = ensure_tower_ws(c, Tc * Leff, t.D);
        if (!rc) rc = ensure(c, c->head16, rup(Tc, 256) * t.D * 2);
        if (!rc) rc = ensure(c, c->i32, (size_t)(Tc + 4) * 4);
        if (rc) return rc;
        int32_t* eot = (int32_t*)c->i32.p;
        // EOT position = argmax over the FULL row (clipnet/model.py:350); must lie inside Leff
        HG_HIP(launch_eot_argmax(ids + (size_t)t0 * L, Tc, L, eot, nullptr, s));
        if (Leff < L) HG_HIP(launch_clamp_eot(eot, Tc, Leff, eot, c->eot_flag, s, c->eot_flag_dev));
        HG_HIP(launch_embed_tokens(ids + (size_t)t0 * L, L, t.tok, t.pos, (float*)c->x.p, Tc, Leff, t.D, t.vocab, s));
        rc = text_tail(c, Tc, Leff, eot, out + (size_t)t0 * t.E, s);
        if (rc) return rc;
    }
    if (Leff < L) HG_HIP(launch_poison_if_flag(out, (size_t)T * t.E, c->eot_flag_dev, s));
    return HG_OK;
}

int hg_encode_text_embeds(hg_ctx* c, const float* prompts, const int32_t* eot_idx, int R, int L, float* out,
                          int trunc, void* stream) {
    if (!c) return HG_ERR_INVALID;
    Text& t = c->text;
    if (!t.loaded) return fail(c, HG_ERR_NOT_LOADED, "hg_load_text has not been called");
    if (R == 0) return HG_OK;
    if (R < 0 || !prompts || !eot_idx || !out || L < 1 || L > t.ctx)
        return fail(c, HG_ERR_INVALID, "bad arguments to encode_text_embeds");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    if (int frc = text_check_flag(c)) return frc;
    const int Leff = (trunc > 0 && trunc < L) ? trunc : L;
    if (Leff < L && c->eot_flag_dev) HG_HIP(hipMemsetAsync(c->eot_flag_dev, 0, 4, s));
    const int chunk = text_chunk_prompts(c, R, Leff);
    for (int r0 = 0; r0 < R; r0 += chunk) {
        const int Rc = (R - r0 < chunk) ? R - r0 : chunk;
        int rc = ensure_tower_ws(c, Rc * Leff, t.D);
        if (!rc) rc = ensure(c, c->head16, rup(Rc, 256) * t.D * 2);
        if (rc) return rc;
        HG_HIP(launch_add_pos(prompts + (size_t)r0 * L * t.D, L, t.pos, (float*)c->x.p, Rc, Leff, t.D, s));
        const int32_t* eot = eot_idx + r0;
        if (Leff < L) {
            rc = ensure(c, c->i32, (size_t)(Rc + 4) * 4);
            if (rc) return rc;
            HG_HIP(launch_clamp_eot(eot, Rc, Leff, (int32_t*)c->i32.p, c->eot_flag, s, c->eot_flag_dev));
            eot = (const int32_t*)c->i32.p;
        }
        rc = text_tail(c, Rc, Leff, eot, out + (size_t)r0 * t.E, s);
        if (rc) return rc;
    }
    if (Leff < L) HG_HIP(launch_poison_if_flag(out, (size_t)R * t.E, c->eot_flag_dev, s));
    return HG_OK;
}

int hg_token_embedding(hg_ctx* c, const int32_t* ids, int n, float* out, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (!c->text.loaded) return fail(c, HG_ERR_NOT_LOADED, "hg_load_text has not been called");
    if (n < 0 || !ids || !out) return fail(c, HG_ERR_INVALID, "bad arguments to token_embedding");
    HG_ON_DEVICE(c);
    HG_HIP(launch_gather_rows(ids, c->text.tok, out, n, c->text.D, c->text.vocab, (hipStream_t)stream));
    return HG_OK;
}

// ---- CoOp-VAE ---------------------------------------------------------------------------------------------
static int vae_fused_rows(const hg_ctx* c, int R) { return fused_item_rows(c, c->opt_vae_fused, R); }
static int generator_rows(hg_ctx* c, Vae& v, const half_t* z16, int R, float* bias, hipStream_t s) {
    // Generator: relu(z W0^T + b0) W2^T + b2  (main_coop_vae.py:282-296)
    half_t* g1 = (half_t*)c->fc.p;
    GemmArgs g{};
    g.A = z16; g.lda = v.dim; g.W = v.g_w0; g.bias = v.g_b0; g.out = g1; g.ldc = v.gh; g.M = R; g.N = v.gh; g.K = v.dim;
    HG_HIP(gemm(c, EPI_BIAS_RELU_F16, g, s));
    g = GemmArgs{};
    g.A = g1; g.lda = v.gh; g.W = v.g_w2; g.bias = v.g_b2; g.out = bias; g.ldc = v.dim; g.M = R; g.N = v.dim; g.K = v.gh;
    HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
    return HG_OK;
}

int hg_vae_forward(hg_ctx* c, int slot, const float* x, const float* eps, int R, float* mean, float* logvar,
                   float* z, float* bias, void* stream) {
    if (!c || slot < 0 || slot >= HG_MAX_SLOTS) return HG_ERR_INVALID;
    Vae& v = c->vae[slot];
    if (!v.enc || (bias && !v.gen)) return fail(c, HG_ERR_NOT_LOADED, "hg_load_vae(slot %d) incomplete", slot);
    if (R == 0) return HG_OK;
    if (R < 0 || !x || !eps) return fail(c, HG_ERR_INVALID, "bad arguments to vae_forward");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const int dim = v.dim;
    // Option vae_fused = 2: the leading rows that fill whole rounds of items (all rows) go through ONE kernel, hidden layers and z on
    // chip (hg_vae_fused.hip).  Default (1): the Encoder stays on the GEMM path - the one kernel computes its hidden layer twice (1 024
    // output columns do not fit a wave's registers) and loses to the GEMMs there: measured 1.94 ms against 1.72 ms for 98 304 rows - and
    // the Generator of those rows runs as the one kernel on the fp16 z the reparameterisation kernel writes (0.86 against 0.92 ms).
    const int Rf = v.wp ? vae_fused_rows(c, R) : 0;
    const bool all_fused = Rf > 0 && c->opt_vae_fused == 2;
    if (all_fused) {
        int rc = ensure(c, c->zpark, vae_fused_park_bytes(Rf));
        if (rc) return rc;
        VaeFusedArgs a{};
        a.x = x; a.eps = eps; a.mean = mean; a.logvar = logvar; a.z = z; a.bias = bias; a.wp = v.wp;
        a.b0e = v.e_b0; a.bml = v.e_bml; a.b0g = v.g_b0; a.b2g = v.g_b2; a.zpark = (half_t*)c->zpark.p;
        a.R = Rf; a.eh = v.eh; a.gh = v.gen ? v.gh : 0; a.mode = bias ? 0 : 1; a.has_enc = true;
        ProfScope ps(c, s, HG_PROF_VAE_FUSED, Rf, bias ? 3 : 2, v.eh);
        HG_HIP(launch_vae_fused(a, s));
    }
    // rows of the call whose Generator runs as the one kernel (hybrid): the chunks below stop at that boundary
    const int Rg = (!all_fused && bias && Rf > 0) ? Rf : 0;
    if (Rg > 0) {      // z of those rows as fp16, for ONE Generator launch behind the chunks (three items per CU instead of three launches)
        int rc = ensure(c, c->zpark, (size_t)rup(Rg, 256) * dim * 2);
        if (rc) return rc;
    }
    for (int r0 = all_fused ? Rf : 0, Rc = 0; r0 < R; r0 += Rc) {
        Rc = chunk_rows(c, R - r0);
        if (r0 < Rg && r0 + Rc > Rg) Rc = Rg - r0;
        const bool gen_fused = r0 < Rg;
        const size_t Rp = rup(Rc, 256);
        int rc = ensure(c, c->h, Rp * dim * 2);
        if (!rc) rc = ensure(c, c->att, Rp * dim * 2);
        if (!rc) rc = ensure(c, c->qkv, Rp * v.eh * 2);
        if (!rc) rc = ensure(c, c->x, Rp * 2 * dim * 4);
        if (!rc && bias && !gen_fused) rc = ensure(c, c->fc, Rp * v.gh * 2);
        if (rc) return rc;
        half_t* x16 = (half_t*)c->h.p;
        const size_t o = (size_t)r0 * dim;
        half_t* z16 = gen_fused ? (half_t*)c->zpark.p + o : (half_t*)c->att.p;
        half_t* h1 = (half_t*)c->qkv.p;
        float* ml = (float*)c->x.p;          // [2][Rp, dim] planes for the halves the caller did not ask for
        float* mean_o = mean ? mean + o : ml;
        float* lv_o = logvar ? logvar + o : ml + Rp * dim;
        HG_HIP(launch_f32_to_f16(x + o, x16, (size_t)Rc * dim, s));
        GemmArgs g{};
        g.A = x16; g.lda = dim; g.W = v.e_w0; g.bias = v.e_b0; g.out = h1; g.ldc = v.eh; g.M = Rc; g.N = v.eh; g.K = dim;
        HG_HIP(gemm(c, EPI_BIAS_RELU_F16, g, s));
        // mean | log_var as ONE N = 2*dim GEMM whose two column halves land directly in the caller's tensors
        g = GemmArgs{};
        g.A = h1; g.lda = v.eh; g.out = mean_o; g.out_hi = lv_o; g.ldc = dim; g.M = Rc; g.N = 2 * dim; g.K = v.eh;
        // (The reparameterisation on the accumulators of a row-interleaved mean | log_var GEMM was built and measured in round 3:
        // bit-identical, one launch and 410 MB less per 100 k rows, 0.5-2 % SLOWER - its 56 partial-line stores per wave cost the
        // GEMM what the HBM-speed reparam kernel costs on its own; commit 2b473ec and earlier carry it.)
        g.W = v.e_wml; g.bias = v.e_bml; g.n_split = dim;
        HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
        HG_HIP(launch_reparam(mean_o, lv_o, eps + o, Rc, dim, z ? z + o : nullptr, z16, dim, s));
        if (bias && !gen_fused) {
            rc = generator_rows(c, v, z16, Rc, bias + o, s);
            if (rc) return rc;
        }
    }
    if (Rg > 0) {
        VaeFusedArgs a{};
        a.x16 = (const half_t*)c->zpark.p; a.bias = bias; a.wp = v.wp; a.b0g = v.g_b0; a.b2g = v.g_b2;
        a.R = Rg; a.eh = v.eh; a.gh = v.gh; a.mode = 2; a.has_enc = true;
        ProfScope ps(c, s, HG_PROF_VAE_FUSED, Rg, 1, v.gh);
        HG_HIP(launch_vae_fused(a, s));
    }
    return HG_OK;
}

int hg_generator(hg_ctx* c, int slot, const float* z, int R, float* bias, void* stream) {
    if (!c || slot < 0 || slot >= HG_MAX_SLOTS) return HG_ERR_INVALID;
    Vae& v = c->vae[slot];
    if (!v.gen) return fail(c, HG_ERR_NOT_LOADED, "generator of slot %d not loaded", slot);
    if (R == 0) return HG_OK;
    if (R < 0 || !z || !bias) return fail(c, HG_ERR_INVALID, "bad arguments to generator");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    const int Rf = v.wp ? vae_fused_rows(c, R) : 0;
    if (Rf > 0) {
        VaeFusedArgs a{};
        a.x = z; a.bias = bias; a.wp = v.wp; a.b0g = v.g_b0; a.b2g = v.g_b2;
        a.R = Rf; a.eh = v.enc ? v.eh : 0; a.gh = v.gh; a.mode = 2; a.has_enc = v.enc;
        ProfScope ps(c, s, HG_PROF_VAE_FUSED, Rf, 1, v.gh);
        HG_HIP(launch_vae_fused(a, s));
    }
    for (int r0 = Rf, Rc = 0; r0 < R; r0 += Rc) {
        Rc = chunk_rows(c, R - r0);
        const size_t Rp = rup(Rc, 256);
        int rc = ensure(c, c->att, Rp * v.dim * 2);
        if (!rc) rc = ensure(c, c->fc, Rp * v.gh * 2);
        if (rc) return rc;
        half_t* z16 = (half_t*)c->att.p;
        HG_HIP(launch_f32_to_f16(z + (size_t)r0 * v.dim, z16, (size_t)Rc * v.dim, s));
        rc = generator_rows(c, v, z16, Rc, bias + (size_t)r0 * v.dim, s);
        if (rc) return rc;
    }
    return HG_OK;
}

int hg_mlp_net(hg_ctx* c, int slot, const float* x, int R, float* out, void* stream) {
    if (!c || slot < 0 || slot >= HG_MAX_SLOTS) return HG_ERR_INVALID;
    Mlp& m = c->mlp[slot];
    if (!m.loaded) return fail(c, HG_ERR_NOT_LOADED, "mlp_net slot %d not loaded", slot);
    if (R == 0) return HG_OK;
    if (R < 0 || !x || !out) return fail(c, HG_ERR_INVALID, "bad arguments to mlp_net");
    hipStream_t s = (hipStream_t)stream;
    HG_ON_DEVICE(c);
    for (int r0 = 0, Rc = 0; r0 < R; r0 += Rc) {
        Rc = chunk_rows(c, R - r0);
        const size_t Rp = rup(Rc, 256);
        int rc = ensure(c, c->h, Rp * m.in * 2);
        if (!rc) rc = ensure(c, c->att, Rp * m.hid * 2);
        if (!rc) rc = ensure(c, c->qkv, Rp * m.hid * 2);
        if (rc) return rc;
        half_t* x16 = (half_t*)c->h.p;
        half_t* a1 = (half_t*)c->att.p;
        half_t* a2 = (half_t*)c->qkv.p;
        HG_HIP(launch_f32_to_f16(x + (size_t)r0 * m.in, x16, (size_t)Rc * m.in, s));
        GemmArgs g{};
        g.A = x16; g.lda = m.in; g.W = m.w0; g.bias = m.b0; g.out = a1; g.ldc = m.hid; g.M = Rc; g.N = m.hid; g.K = m.in;
        HG_HIP(gemm(c, EPI_BIAS_RELU_F16, g, s));
        g = GemmArgs{};
        g.A = a1; g.lda = m.hid; g.W = m.w2; g.bias = m.b2; g.out = a2; g.ldc = m.hid; g.M = Rc; g.N = m.hid; g.K = m.hid;
        HG_HIP(gemm(c, EPI_BIAS_RELU_F16, g, s));
        g = GemmArgs{};
        g.A = a2; g.lda = m.hid; g.W = m.w4; g.bias = m.b4; g.out = out + (size_t)r0 * m.out; g.ldc = m.out; g.M = Rc; g.N = m.out; g.K = m.hid;
        HG_HIP(gemm(c, EPI_BIAS_F32, g, s));
    }
    return HG_OK;
}

int hg_assemble_prompts(hg_ctx* c, const float* prefix, const float* suffix, const float* ctx, const float* bias,
                        const int32_t* target, int R, int C, int L, int n_ctx, int D, float* prompts, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (!prefix || !suffix || !ctx || !bias || !target || !prompts || R < 0 || C <= 0)
        return fail(c, HG_ERR_INVALID, "bad arguments to assemble_prompts");
    HG_ON_DEVICE(c);
    HG_HIP(launch_assemble_prompts(prefix, suffix, ctx, bias, target, R, C, L, n_ctx, D, prompts, (hipStream_t)stream));
    return HG_OK;
}

int hg_l2_normalize(hg_ctx* c, const float* x, int R, int D, float* out, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (!x || !out || R < 0 || D <= 0) return fail(c, HG_ERR_INVALID, "bad arguments to l2_normalize");
    HG_ON_DEVICE(c);
    HG_HIP(launch_l2_normalize(x, out, R, D, (hipStream_t)stream));
    return HG_OK;
}

int hg_vae_loss(hg_ctx* c, const float* recon, const float* x, const float* mean, const float* logvar, int R,
                int D, float* loss, void* stream) {
    if (!c) return HG_ERR_INVALID;
    if (!recon || !x || !mean || !logvar || !loss || R <= 0) return fail(c, HG_ERR_INVALID, "bad arguments to vae_loss");
    HG_ON_DEVICE(c);
    HG_HIP(launch_vae_loss(recon, x, mean, logvar, R, D, loss, (hipStream_t)stream));
    return HG_OK;
}

}  // extern "C"
