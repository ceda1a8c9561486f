// Instance adapter of the detector's CLIP (variant C) — CLIP_models_adapter_prior2.py:142-203 and the
// post-norm TransformerDecoderLayer it embeds (:27-72).  SURVEY.md §2.2 K9.
//
// Per block:  x += scale * up_proj( DecoderLayer( relu(down_proj(x)), memory ) )
//   memory = prior tokens [B,N,64] (key_padding_mask) when a prior is given, else `down` itself.
// down_proj (K=768 -> 64, padded to a 128-wide tile) and up_proj (K=64 -> 768, fused scale+residual
// epilogue) run on the MFMA GEMM.  The 64-wide decoder layer in between (cross attention with 2 heads
// of 32, FFN 64-128-64, two LayerNorms) is 1.7 % of a block's FLOPs and runs in fp32 on the VALU:
// one lane per token, the token's activations in a private LDS row, weights fetched with wave-uniform
// (scalar-cache) loads so every FMA takes an SGPR weight operand.
#include <stdlib.h>

#include "hg_kernels.h"

namespace hg {

static constexpr int AD = 64;         // bottleneck width
static constexpr int ROWP = 2 * AD + 1;   // LDS row pitch (floats): conflict-free per-lane rows

// acc[o] += sum_i WT[i][o] * row[i],  WT wave-uniform [n_in][N_OUT]
template <int N_OUT>
__device__ __forceinline__ void matvec(float (&acc)[N_OUT], const float* __restrict__ WT, const float* row, int n_in) {
    for (int i = 0; i < n_in; ++i) {
        const float xi = row[i];
        const float* w = WT + (size_t)i * N_OUT;
#pragma unroll
        for (int o = 0; o < N_OUT; ++o) acc[o] = fmaf(w[o], xi, acc[o]);
    }
}

__device__ __forceinline__ void layer_norm64(float (&v)[AD], const float* __restrict__ w, const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int o = 0; o < AD; ++o) s += v[o];
    const float mean = s * (1.0f / AD);
    float q = 0.f;
#pragma unroll
    for (int o = 0; o < AD; ++o) {
        const float d = v[o] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(q * (1.0f / AD) + 1e-5f);
#pragma unroll
    for (int o = 0; o < AD; ++o) v[o] = (v[o] - mean) * rstd * w[o] + b[o];
}

// K/V projections of the memory tokens: mem [n_rows, ld] fp32 -> kv [n_rows, 2, 64]
__global__ __launch_bounds__(64) void adapter_kv_kernel(const float* __restrict__ mem, int ld, int n_rows,
                                                        const float* __restrict__ WkT, const float* __restrict__ bk,
                                                        const float* __restrict__ WvT, const float* __restrict__ bv,
                                                        float* __restrict__ kv) {
    __shared__ float rows[64][AD + 1];
    const int lane = threadIdx.x;
    const int r = blockIdx.x * 64 + lane;
    const int rr = r < n_rows ? r : n_rows - 1;
#pragma unroll
    for (int i = 0; i < AD; ++i) rows[lane][i] = mem[(size_t)rr * ld + i];
    float k[AD], v[AD];
#pragma unroll
    for (int o = 0; o < AD; ++o) {
        k[o] = bk[o];
        v[o] = bv[o];
    }
    matvec<AD>(k, WkT, rows[lane], AD);
    matvec<AD>(v, WvT, rows[lane], AD);
    if (r < n_rows) {
        f32x4* dst = reinterpret_cast<f32x4*>(kv + (size_t)r * 2 * AD);
#pragma unroll
        for (int o = 0; o < AD; o += 4) {
            dst[o / 4] = f32x4{k[o], k[o + 1], k[o + 2], k[o + 3]};
            dst[AD / 4 + o / 4] = f32x4{v[o], v[o + 1], v[o + 2], v[o + 3]};
        }
    }
}

struct DecoderPtrs {
    const float *WqT, *bq, *WoT, *bo, *norms, *W1T, *b1, *W2T_b2;
};

__global__ __launch_bounds__(64) void adapter_decoder_kernel(const float* __restrict__ down, int ld_down,
                                                             DecoderPtrs P, const float* __restrict__ kv,
                                                             const uint8_t* __restrict__ mask, int L, int Nmem,
                                                             int chunks, half_t* __restrict__ out16, int ld16) {
    __shared__ float rows[64][ROWP];
    const int lane = threadIdx.x;
    const int seq = blockIdx.x / chunks, ch = blockIdx.x - seq * chunks;
    const int t = ch * 64 + lane;
    const bool valid = t < L;
    const size_t m = (size_t)seq * L + (valid ? t : L - 1);
    float* row = rows[lane];

    float tgt[AD];
#pragma unroll
    for (int o = 0; o < AD; o += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(down + m * ld_down + o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tgt[o + e] = v[e];
            row[o + e] = v[e];
        }
    }
    // ---- q = Wq tgt + bq
    float q[AD];
#pragma unroll
    for (int o = 0; o < AD; ++o) q[o] = P.bq[o];
    matvec<AD>(q, P.WqT, row, AD);
    // ---- cross attention, 2 heads of 32, scale 32^-0.5, key_padding_mask (True = ignore)
    const float scale = 0.17677669529663687f;
    const float* kvs = kv + (size_t)seq * Nmem * 2 * AD;
    const uint8_t* mk = mask ? mask + (size_t)seq * Nmem : nullptr;
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
        float mrun = -INFINITY, lsum = 0.f;
        float acc[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) acc[d] = 0.f;
        for (int n = 0; n < Nmem; ++n) {
            if (mk && mk[n]) continue;   // wave-uniform
            const float* kr = kvs + (size_t)n * 2 * AD + hd * 32;
            const float* vr = kr + AD;
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 32; ++d) s = fmaf(q[hd * 32 + d], kr[d], s);
            s *= scale;
            const float mnew = fmaxf(mrun, s);
            const float alpha = __expf(mrun - mnew);
            const float p = __expf(s - mnew);
            lsum = lsum * alpha + p;
#pragma unroll
            for (int d = 0; d < 32; ++d) acc[d] = fmaf(p, vr[d], acc[d] * alpha);
            mrun = mnew;
        }
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int d = 0; d < 32; ++d) row[hd * 32 + d] = acc[d] * inv;
    }
    // ---- out_proj, residual, norm2
    {
        float o2[AD];
#pragma unroll
        for (int o = 0; o < AD; ++o) o2[o] = P.bo[o];
        matvec<AD>(o2, P.WoT, row, AD);
#pragma unroll
        for (int o = 0; o < AD; ++o) tgt[o] += o2[o];
    }
    layer_norm64(tgt, P.norms, P.norms + AD);
#pragma unroll
    for (int o = 0; o < AD; ++o) row[o] = tgt[o];
    // ---- FFN 64 -> 128 (relu) -> 64, residual, norm3
    {
        float hid[2 * AD];
#pragma unroll
        for (int o = 0; o < 2 * AD; ++o) hid[o] = P.b1[o];
        matvec<2 * AD>(hid, P.W1T, row, AD);
#pragma unroll
        for (int o = 0; o < 2 * AD; ++o) row[o] = fmaxf(hid[o], 0.f);
    }
    {
        float f[AD];
        const float* b2 = P.W2T_b2 + (size_t)2 * AD * AD;
#pragma unroll
        for (int o = 0; o < AD; ++o) f[o] = b2[o];
        matvec<AD>(f, P.W2T_b2, row, 2 * AD);
#pragma unroll
        for (int o = 0; o < AD; ++o) tgt[o] += f[o];
    }
    layer_norm64(tgt, P.norms + 2 * AD, P.norms + 3 * AD);
    if (valid) {
        half_t* dst = out16 + (size_t)m * ld16;
#pragma unroll
        for (int o = 0; o < AD; o += 8) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (half_t)tgt[o + e];
            *reinterpret_cast<half8*>(dst + o) = h;
        }
    }
}

// =====================================================================================================================
// MFMA path of the same decoder layer: one workgroup (4 waves, 64 tokens each) per sequence.
//
// Every 64-wide linear runs as D = W . X^T on v_mfma_f32_16x16x32_f16 with the weight tile as the A operand and 16
// token rows as the B operand, so an activation [64 tokens x 16g..] sits in registers as t[f][g] (f32x4): token
// 16f + (lane & 15), features 16g + 4(lane >> 4) + r - the same "a lane holds 4 consecutive columns of one row" layout as
// the GEMM kernels.  Between stages the activation goes through the wave's private LDS rows (fp16, 8-byte stores,
// 16-byte fragment reads); weights are read from global memory (64 KB per layer, L2-resident) straight into fragments.
// Attention: scores S[key][token] = K_h . Q_h^T per head (K = 32 features = one MFMA k-step), softmax over the keys of a
// token across the lane's registers and its four lane rows, and P . V with the probabilities used as the B operand
// exactly where the score accumulators left them (key order inside a 32-key block permuted; V^T is stored in LDS in
// that order).  Residual stream of the layer, LayerNorm statistics and softmax stay in fp32.
struct DecW16 {
    const half_t *Wq, *Wk, *Wv, *Wo, *W1, *W2;      // fp16 [out][in]: 64x64 (x4), 128x64, 64x128
    const float *bq, *bk, *bv, *bo, *b1, *b2, *norms;   // norms = {norm2.w, norm2.b, norm3.w, norm3.b}
};

static constexpr int XP = 136;      // halfs per row of a wave's activation buffer (64 or 128 features + pad)
static constexpr int KP = 72;       // halfs per row of K [key][64]
static constexpr int NKMAX = 224;   // keys: 14 tiles of 16 (L <= 224)
static constexpr int VP = NKMAX + 8;   // halfs per row of V^T [feature][key slot]

typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max_rows(float x) {      // max over the four 16-lane rows
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float sum_rows4(float x) {
    x += __shfl_xor(x, 16, 64);
    return x + __shfl_xor(x, 32, 64);
}

// t[f][g] (+)= W[16g.., :] . X[16f.., :]^T over K = 32 * KS features; W global fp16 [.][ldw], X = LDS rows of pitch XP
template <int G, int KS, int F>
__device__ __forceinline__ void linear_T(f32x4 (&t)[F][G], const half_t* __restrict__ W, int ldw, const half_t* X, int lane) {
    const int r = lane & 15, q = lane >> 4;
    half8 xf[F][KS];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[f][ks] = *reinterpret_cast<const half8*>(X + (16 * f + r) * XP + 32 * ks + 8 * q);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const half8 wf = *reinterpret_cast<const half8*>(W + (size_t)(16 * g + r) * ldw + 32 * ks + 8 * q);
#pragma unroll
            for (int f = 0; f < F; ++f) t[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[f][ks], t[f][g], 0, 0, 0);
        }
}
template <int G, int F>
__device__ __forceinline__ void init_bias(f32x4 (&t)[F][G], const float* __restrict__ b, int lane) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + 16 * g + 4 * (lane >> 4));
#pragma unroll
        for (int f = 0; f < F; ++f) t[f][g] = bv;
    }
}
template <int G, int F>
__device__ __forceinline__ void store_T(const f32x4 (&t)[F][G], half_t* X, int lane) {
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)t[f][g][e];
            *reinterpret_cast<half4*>(X + (16 * f + r) * XP + 16 * g + 4 * q) = h;
        }
}
// LayerNorm over the 64 features of every token (eps 1e-5, biased variance), in place; AFFINE = false: normalised values only
template <bool AFFINE = true, int F>
__device__ __forceinline__ void layer_norm_T(f32x4 (&t)[F][4], const float* __restrict__ w, const float* __restrict__ b, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) s += (t[f][g][0] + t[f][g][1]) + (t[f][g][2] + t[f][g][3]);
        const float mean = sum_rows4(s) * (1.0f / 64.0f);
        float v = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = t[f][g][e] - mean;
                v = fmaf(d, d, v);
            }
        const float rstd = 1.0f / sqrtf(sum_rows4(v) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (AFFINE) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 16 * g + 4 * q), bv = *reinterpret_cast<const f32x4*>(b + 16 * g + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[f][g][e] = (t[f][g][e] - mean) * rstd * wv[e] + bv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) t[f][g][e] = (t[f][g][e] - mean) * rstd;
            }
        }
    }
}

// key index -> slot inside the V^T rows: within a block of 32 keys, key 16*kt + 4*q + r sits at 8*q + 4*kt + r, which is
// where the score accumulators of key tiles (2b, 2b+1) put it when they are packed as the P.V B operand
__device__ __forceinline__ int key_slot(int key) {
    const int w = key & 31;
    return (key & ~31) + 8 * ((w & 15) >> 2) + 4 * (w >> 4) + (w & 3);
}

// F = 16-token tiles per wave: the workgroup has ceil(L / 16F) waves (F = 2: seven waves of 32 tokens at L = 197 - two
// waves per SIMD hide each other's LDS round trips and weight fetches; F = 4: four waves of 64)
// DN: down_proj runs here as well (AdapterDownDev): the workgroup owns its sequence's rows, so relu(W_down x + b) and the
// cross-term columns x16 Q never travel through HBM as fp32 (26 MB each way per layer at B = 256) and one launch goes
template <bool SELF, int F, bool DN>
// `down` is not __restrict__: a chained layer (adapter_num_layers > 1) writes chain32 == down in place
__global__ __launch_bounds__(F == 2 ? 512 : 256) void adapter_decoder_mfma(const float* down, int ld_down, DecW16 W,
                                                            const float* __restrict__ priors, const uint8_t* __restrict__ mask,
                                                            int L, int N, half_t* __restrict__ out16, float* chain32, int ld16,
                                                            AdapterFoldDev FD, AdapterDownDev DD) {
    extern __shared__ __attribute__((aligned(16))) char smem_ad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int seq = blockIdx.x;
    const int nkeys = SELF ? L : N;
    const int nkt = (nkeys + 15) >> 4;                  // key tiles of 16
    const int nkb = (nkt + 1) >> 1;                     // key blocks of 32
    half_t* Xw = reinterpret_cast<half_t*>(smem_ad) + wave * 16 * F * XP;        // this wave's activation rows
    half_t* Ks = reinterpret_cast<half_t*>(smem_ad) + 4 * 64 * XP;           // K [key][KP]
    half_t* VT = Ks + NKMAX * KP;                                            // V^T [feature][VP]
    half_t* Mem = VT + 64 * VP;                                              // prior tokens fp16 [32][XP] (prior case)

    // ---- this wave's 64 tokens: fp32 residual of the layer in registers, fp16 copy in LDS
    f32x4 tgt[F][4];
    f32x4 wq[DN ? F : 1][4];      // DN: x16 Q of this wave's tokens (the cross term of the folded statistics)
    int tok[F];
#pragma unroll
    for (int f = 0; f < F; ++f) tok[f] = wave * 16 * F + 16 * f + r;
    if constexpr (DN) {
        // [relu(down_proj) | x16 Q] = [x16 + mu] W2^T + b over K = D: the 128 x 64 weight tile of a K-tile is staged through LDS
        // (the K / V area is idle until emit_kv) for all waves, double buffered, the next tile's global loads in flight under the
        // current tile's MFMAs; a wave's own rows come straight from global memory as MFMA fragments, one K-tile ahead.
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int WP = 72;                       // halfs per staged weight row (64 + pad)
        half_t* Wst = Ks;
        const int nthr = (int)blockDim.x, nkd = DD.K >> 6;
        size_t mrow[F], xrow[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            mrow[f] = (size_t)seq * L + (tok[f] < L ? tok[f] : L - 1);
            xrow[f] = mrow[f] * DD.ldx;
        }
        f32x4 t8[F][8];
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int g = 0; g < 8; ++g) t8[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        // 1024 pieces of 16 B per weight tile: three per thread (>= 384 threads).  Two register sets (tiles kt+1, kt+2 in flight),
        // two LDS buffers, two fragment sets of the wave's own rows: every global load is issued two K-tiles before its use
        u32x4 stA[3], stB[3];
        auto ld_w = [&](u32x4 (&st)[3], int kt) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pc = tid + j * nthr;
                if (pc < 1024) st[j] = *reinterpret_cast<const u32x4*>(DD.w + (size_t)(pc >> 3) * DD.K + kt * 64 + (pc & 7) * 8);
            }
        };
        auto st_w = [&](const u32x4 (&st)[3], int buf) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int pc = tid + j * nthr;
                if (pc < 1024) *reinterpret_cast<u32x4*>(Wst + buf * 128 * WP + (pc >> 3) * WP + (pc & 7) * 8) = st[j];
            }
        };
        half8 x0[F][2], x1[F][2];
        auto ld_x = [&](half8 (&x)[F][2], int kt) {
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) x[f][ks] = *reinterpret_cast<const half8*>(DD.x16 + xrow[f] + kt * 64 + 32 * ks + 8 * q);
        };
        auto compute = [&](int buf, const half8 (&x)[F][2]) {
            const half_t* wt = Wst + buf * 128 * WP;
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const half8 wf = *reinterpret_cast<const half8*>(wt + (16 * g + r) * WP + 32 * ks + 8 * q);
#pragma unroll
                    for (int f = 0; f < F; ++f) t8[f][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, x[f][ks], t8[f][g], 0, 0, 0);
                }
        };
        ld_w(stA, 0);
        ld_x(x0, 0);
        ld_x(x1, 1);
        st_w(stA, 0);
        ld_w(stA, 1);
        if (2 < nkd) ld_w(stB, 2);
        for (int kt = 0; kt < nkd; kt += 2) {        // (K / 64 is even: the launcher checks K % 128)
            __syncthreads();                         // tile kt is in buffer 0; every wave is done with tile kt - 1 (buffer 1)
            compute(0, x0);
            st_w(stA, 1);                            // tile kt + 1
            if (kt + 3 < nkd) ld_w(stA, kt + 3);
            if (kt + 2 < nkd) ld_x(x0, kt + 2);
            __syncthreads();
            compute(1, x1);
            if (kt + 2 < nkd) st_w(stB, 0);
            if (kt + 4 < nkd) ld_w(stB, kt + 4);
            if (kt + 3 < nkd) ld_x(x1, kt + 3);
        }
        __syncthreads();                             // the staging area becomes K / V
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float mu = DD.muc[mrow[f]];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 cs4 = *reinterpret_cast<const f32x4*>(DD.cs + 16 * g + 4 * q), b4 = *reinterpret_cast<const f32x4*>(DD.b + 16 * g + 4 * q);
                f32x4 v = t8[f][g] + cs4 * mu + b4;  // W (x16 + mu) + b = acc + mu * cs + b
                if (g < 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    tgt[f][g] = v;
                } else {
                    wq[f][g - 4] = v;
                }
            }
            if (chain32 && tok[f] < L) {             // chained layers: the last one reads x16 Q from the fp32 buffer
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(chain32 + ((size_t)seq * L + tok[f]) * ld_down + 64 + 16 * g + 4 * q) = wq[f][g];
            }
        }
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const size_t m = (size_t)seq * L + (tok[f] < L ? tok[f] : L - 1);
#pragma unroll
            for (int g = 0; g < 4; ++g) tgt[f][g] = *reinterpret_cast<const f32x4*>(down + m * ld_down + 16 * g + 4 * q);
        }
    }
    store_T<4>(tgt, Xw, lane);
    // ---- K and V of the memory tokens -> LDS (K row-major, V transposed with permuted key slots)
    auto emit_kv = [&](const half_t* X, int key0) {       // 16 F memory rows in X -> keys key0 ..
        f32x4 kk[F][4], vv[F][4];
        init_bias<4>(kk, W.bk, lane);
        init_bias<4>(vv, W.bv, lane);
        linear_T<4, 2>(kk, W.Wk, 64, X, lane);
        linear_T<4, 2>(vv, W.Wv, 64, X, lane);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int key = key0 + 16 * f + r;
            if (key < NKMAX) {
                const int slot = key_slot(key);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (half_t)kk[f][g][e];
                    *reinterpret_cast<half4*>(Ks + key * KP + 16 * g + 4 * q) = h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) VT[(16 * g + 4 * q + e) * VP + slot] = (half_t)vv[f][g][e];
                }
            }
        }
    };
    if constexpr (SELF) {
        emit_kv(Xw, wave * 16 * F);                            // memory = the sequence's own tokens
        __syncthreads();
    } else {
        // prior tokens [N <= 32][64] fp32 -> fp16 rows (rows >= N zero), then wave 0 projects them
        for (int i = tid; i < 32 * 16; i += (int)blockDim.x) {
            const int row = i >> 4, c4 = (i & 15) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < N) v = *reinterpret_cast<const f32x4*>(priors + ((size_t)seq * N + row) * 64 + c4);
            half4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
            *reinterpret_cast<half4*>(Mem + row * XP + c4) = h;
        }
        for (int i = tid; i < 32 * 16; i += (int)blockDim.x) {          // rows 32..63 of the fragment reads: zeros
            *reinterpret_cast<half4*>(Mem + (32 + (i >> 4)) * XP + (i & 15) * 4) = half4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        }
        __syncthreads();
        if (wave * 16 * F < 32) emit_kv(Mem + wave * 16 * F * XP, wave * 16 * F);      // the 32 prior rows
        __syncthreads();
    }
    // masked keys as a bit set (wave-uniform): bit k of word k >> 5
    unsigned mbits[NKMAX / 32];
#pragma unroll
    for (int b = 0; b < NKMAX / 32; ++b) {
        unsigned m = 0u;
        if (b < nkb) {
            const int key = 32 * b + (lane & 31);
            bool dead = key >= nkeys;
            if (!SELF && !dead && mask) dead = mask[(size_t)seq * N + key] != 0;
            const unsigned long long bal = __ballot(dead);
            m = (unsigned)bal;                               // lanes 0..31 carry keys 32b .. 32b+31
        }
        mbits[b] = __builtin_amdgcn_readfirstlane(m);
    }

    // ---- q = (Wq x + bq) / sqrt(32)
    f32x4 t[F][4];
    init_bias<4>(t, W.bq, lane);
    linear_T<4, 2>(t, W.Wq, 64, Xw, lane);
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) t[f][g] *= 0.17677669529663687f;
    store_T<4>(t, Xw, lane);                                  // (the fp16 copy of the input is no longer needed)
    // ---- cross attention, 2 heads of 32
    f32x4 att[F][4];
#pragma unroll
    for (int hd = 0; hd < 2; ++hd) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const half8 qf = *reinterpret_cast<const half8*>(Xw + (16 * f + r) * XP + 32 * hd + 8 * q);
            f32x4 o0 = f32x4{0.f, 0.f, 0.f, 0.f}, o1 = o0;
            // pass 1: scores of every key tile -> running maximum (scores are recomputed in pass 2: one MFMA per tile)
            float mx = -INFINITY;
            for (int kt = 0; kt < nkt; ++kt) {
                const half8 kf = *reinterpret_cast<const half8*>(Ks + (16 * kt + r) * KP + 32 * hd + 8 * q);
                f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const unsigned mb = mbits[kt >> 1] >> (16 * (kt & 1) + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (!((mb >> e) & 1u)) mx = fmaxf(mx, s[e]);
            }
            mx = max_rows(mx);
            float lsum = 0.f;
            for (int b = 0; b < nkb; ++b) {
                half8 pf;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
                    const int kt = 2 * b + k2;
                    f32x4 s = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    unsigned mb = 0xFu;
                    if (kt < nkt) {
                        const half8 kf = *reinterpret_cast<const half8*>(Ks + (16 * kt + r) * KP + 32 * hd + 8 * q);
                        s = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        mb = mbits[b] >> (16 * k2 + 4 * q);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pe = ((mb >> e) & 1u) ? 0.f : __expf(s[e] - mx);
                        lsum += pe;
                        pf[4 * k2 + e] = (half_t)pe;
                    }
                }
                const half8 v0 = *reinterpret_cast<const half8*>(VT + (32 * hd + r) * VP + 32 * b + 8 * q);
                const half8 v1 = *reinterpret_cast<const half8*>(VT + (32 * hd + 16 + r) * VP + 32 * b + 8 * q);
                o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, pf, o1, 0, 0, 0);
            }
            const float inv = 1.0f / sum_rows4(lsum);
            att[f][2 * hd] = o0 * inv;
            att[f][2 * hd + 1] = o1 * inv;
        }
    }
    // ---- out_proj, residual, norm2
    store_T<4>(att, Xw, lane);
    init_bias<4>(t, W.bo, lane);
    linear_T<4, 2>(t, W.Wo, 64, Xw, lane);
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) tgt[f][g] += t[f][g];
    layer_norm_T(tgt, W.norms, W.norms + 64, lane);
    // ---- FFN 64 -> 128 (relu) -> 64, residual, norm3
    store_T<4>(tgt, Xw, lane);
    {
        f32x4 hid[F][8];
        init_bias<8>(hid, W.b1, lane);
        linear_T<8, 2>(hid, W.W1, 64, Xw, lane);
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) hid[f][g][e] = fmaxf(hid[f][g][e], 0.f);
        store_T<8>(hid, Xw, lane);
    }
    init_bias<4>(t, W.b2, lane);
    linear_T<4, 4>(t, W.W2, 128, Xw, lane);
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) tgt[f][g] += t[f][g];
    if (FD.mr) {
        // Folded adapter (hg_elem.hip adapter_q_kernel): the layer hands on e = [z_0 .. z_62, 1] (z = norm3 without its affine
        // part) - the block's QKV and out-proj GEMMs take the update a = Q e as 64 more K columns - and the statistics ln_1
        // needs of y = x + a, from those of x and three 64-wide products:
        //   sum_j a_j = e . qm,   sum_j (x_j - c) a_j = e . w'  (w' = x16 Q: columns 64.. of the down_proj GEMM),   sum_j a_j^2 = e^T G e
        layer_norm_T<false>(tgt, nullptr, nullptr, lane);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            if (q == 3) tgt[f][3][3] = 1.0f;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) tgt[f][g][e] = (float)(half_t)tgt[f][g][e];      // as the GEMMs will read it
        }
        store_T<4>(tgt, Xw, lane);
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) t[f][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        linear_T<4, 2>(t, FD.g16, 64, Xw, lane);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const size_t m = (size_t)seq * L + (tok[f] < L ? tok[f] : L - 1);
            float sa = 0.f, cr = 0.f, qd = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 qmv = *reinterpret_cast<const f32x4*>(FD.qm + 16 * g + 4 * q);
                f32x4 wv;
                if constexpr (DN) wv = wq[f][g];
                else wv = *reinterpret_cast<const f32x4*>(down + m * ld_down + 64 + 16 * g + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sa = fmaf(tgt[f][g][e], qmv[e], sa);
                    cr = fmaf(tgt[f][g][e], wv[e], cr);
                    qd = fmaf(tgt[f][g][e], t[f][g][e], qd);
                }
            }
            sa = sum_rows4(sa);
            cr = sum_rows4(cr);
            qd = sum_rows4(qd);
            if (q == 0 && tok[f] < L) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2* mrp = reinterpret_cast<f32x2*>(FD.mr + 2 * m);
                const f32x2 old = *mrp;                          // (mean_x - c, rstd_x)
                const float var_x = 1.0f / (old[1] * old[1]) - 1e-5f;
                const float dv = (2.0f * (cr - old[0] * sa) + (qd - sa * sa * FD.inv_D)) * FD.inv_D;
                const float var_y = fmaxf(var_x + dv, 0.f);
                *mrp = f32x2{old[0] + sa * FD.inv_D, 1.0f / sqrtf(var_y + 1e-5f)};
            }
        }
    } else {
        layer_norm_T(tgt, W.norms + 128, W.norms + 192, lane);
    }
    if (chain32) {      // adapter_num_layers > 1: fp32, in the layout the next layer of the chain reads (rows of ld_down)
#pragma unroll
        for (int f = 0; f < F; ++f)
            if (tok[f] < L) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(chain32 + ((size_t)seq * L + tok[f]) * ld_down + 16 * g + 4 * q) = tgt[f][g];
            }
        return;
    }
#pragma unroll
    for (int f = 0; f < F; ++f)
        if (tok[f] < L) {
            half_t* dst = out16 + ((size_t)seq * L + tok[f]) * ld16;
            half_t* dst2 = FD.e2 ? FD.e2 + ((size_t)seq * L + tok[f]) * ld16 : nullptr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)tgt[f][g][e];
                *reinterpret_cast<half4*>(dst + 16 * g + 4 * q) = h;
                if (dst2) *reinterpret_cast<half4*>(dst2 + 16 * g + 4 * q) = h;
            }
        }
}

bool adapter_decoder_mfma_ok(const AdapterDev& ad, bool priors, int L, int N);
// down_proj inside the decoder kernel: the MFMA path (32 tokens per wave) with at least six waves (L >= 161); shorter sequences
// run down_proj as its own GEMM
bool adapter_decoder_fused_down_ok(const AdapterDev& ad, bool priors, int L, int N) {
    return adapter_decoder_mfma_ok(ad, priors, L, N) && 64 * ((L + 31) / 32) >= 384;
}
// the MFMA decoder serves sequences of at most NKMAX tokens and at most 32 prior tokens; anything else runs the fp32
// one-lane-per-token kernels below (in -DHG_EXPERIMENTS builds HG_ADAPTER_MFMA=0 forces those for A/B timing)
bool adapter_decoder_mfma_ok(const AdapterDev& ad, bool priors, int L, int N) {
#ifdef HG_EXPERIMENTS
    static const bool on = []() { const char* e = getenv("HG_ADAPTER_MFMA"); return !(e && e[0] == '0'); }();
    if (!on) return false;
#endif
    return ad.w16[priors ? 0 : 1][0] && L <= NKMAX && (priors ? N <= 32 : true);
}

// down32 [M,128] fp32 (cols 0..63 = relu(down_proj(x))) -> out16 [M,64] fp16 = decoder layer output
hipError_t launch_adapter_decoder(const float* down32, const AdapterDev& ad, const float* priors,
                                  const uint8_t* mask, int B, int L, int N, float* kv, half_t* out16,
                                  hipStream_t s, float* chain32, int ld16, const AdapterFoldDev* fold,
                                  const AdapterDownDev* dn) {
    if (ld16 < AD || ld16 % 8) return hipErrorInvalidValue;
    AdapterFoldDev F{};
    if (fold && !chain32) F = *fold;
    AdapterDownDev DDv{};
    if (dn) DDv = *dn;
    const int which = priors ? 0 : 1;            // mhsa_layers.0 (prior) vs mhsa (self)
    const float* const* dl = ad.dl[which];
    const int Nmem = priors ? N : L;
    const int n_rows = B * Nmem;
    if (n_rows <= 0) return hipSuccess;
    if (adapter_decoder_mfma_ok(ad, priors != nullptr, L, N)) {
        const half_t* const* w = ad.w16[which];
        DecW16 Wd{w[0], w[1], w[2], w[3], w[4], w[5], dl[3], dl[4], dl[5], dl[7], dl[10], dl[11] + (size_t)2 * AD * AD, dl[8]};
        const int lds = (4 * 64 * XP + NKMAX * KP + 64 * VP + 64 * XP) * 2;
        static bool attr_set_d[HG_MAX_DEVICES] = {};
        bool& attr_set = attr_set_d[current_device_index()];
        if (!attr_set) {
            const void* fns[4] = {reinterpret_cast<const void*>(&adapter_decoder_mfma<true, 2, false>),
                                  reinterpret_cast<const void*>(&adapter_decoder_mfma<false, 2, false>),
                                  reinterpret_cast<const void*>(&adapter_decoder_mfma<true, 2, true>),
                                  reinterpret_cast<const void*>(&adapter_decoder_mfma<false, 2, true>)};
            for (const void* fn : fns) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) return e;
            }
            attr_set = true;
        }
        // 16 F tokens per wave, F = 2: seven waves at L = 197 (prior 30 us, self 46 us per layer at B = 256; four waves of 64
        // tokens took 39 / 67 us, thirteen of 16 tokens 37 / 46 us)
        constexpr int tiles = 2;
        const int nkeys = priors ? N : L;
        const unsigned threads = 64u * (unsigned)((L + 16 * tiles - 1) / (16 * tiles));
#define HG_DEC_LAUNCH(SELF_, F_)                                                                                                     \
    hipLaunchKernelGGL((adapter_decoder_mfma<SELF_, F_, false>), dim3(B), dim3(threads), lds, s, down32, 128, Wd, priors, mask, L, \
                       nkeys, out16, chain32, ld16, F, DDv)
        if (dn) {      // down_proj fused: 32 tokens per wave, >= 384 threads (adapter_decoder_fused_down_ok)
            if (threads < 384 || !dn->x16 || dn->K % 128 || dn->K < 256 || dn->ldx % 8) return hipErrorInvalidValue;
            if (priors)
                hipLaunchKernelGGL((adapter_decoder_mfma<false, 2, true>), dim3(B), dim3(threads), lds, s, down32, 128, Wd, priors, mask, L,
                                   nkeys, out16, chain32, ld16, F, DDv);
            else
                hipLaunchKernelGGL((adapter_decoder_mfma<true, 2, true>), dim3(B), dim3(threads), lds, s, down32, 128, Wd, priors, mask, L,
                                   nkeys, out16, chain32, ld16, F, DDv);
        } else { if (priors) HG_DEC_LAUNCH(false, 2); else HG_DEC_LAUNCH(true, 2); }
#undef HG_DEC_LAUNCH
        return hipGetLastError();
    }
    if (chain32 || F.mr) return hipErrorInvalidValue;      // chained layers and the folded statistics exist only on the MFMA path
    hipLaunchKernelGGL(adapter_kv_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, s, priors ? priors : down32,
                       priors ? AD : 128, n_rows, dl[1], dl[4], dl[2], dl[5], kv);
    DecoderPtrs P{dl[0], dl[3], dl[6], dl[7], dl[8], dl[9], dl[10], dl[11]};
    const int chunks = (L + 63) / 64;
    hipLaunchKernelGGL(adapter_decoder_kernel, dim3(B * chunks), dim3(64), 0, s, down32, 128, P, kv,
                       priors ? mask : (const uint8_t*)nullptr, L, Nmem, chunks, out16, ld16);
    return hipGetLastError();
}

}  // namespace hg
