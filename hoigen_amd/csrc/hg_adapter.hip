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
                                                             int chunks, half_t* __restrict__ out16) {
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
        half_t* dst = out16 + m * AD;
#pragma unroll
        for (int o = 0; o < AD; o += 8) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (half_t)tgt[o + e];
            *reinterpret_cast<half8*>(dst + o) = h;
        }
    }
}

// down32 [M,128] fp32 (cols 0..63 = relu(down_proj(x))) -> out16 [M,64] fp16 = decoder layer output
hipError_t launch_adapter_decoder(const float* down32, const AdapterDev& ad, const float* priors,
                                  const uint8_t* mask, int B, int L, int N, float* kv, half_t* out16,
                                  hipStream_t s) {
    const int which = priors ? 0 : 1;            // mhsa_layers.0 (prior) vs mhsa (self)
    const float* const* dl = ad.dl[which];
    const int Nmem = priors ? N : L;
    const int n_rows = B * Nmem;
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(adapter_kv_kernel, dim3((n_rows + 63) / 64), dim3(64), 0, s, priors ? priors : down32,
                       priors ? AD : 128, n_rows, dl[1], dl[4], dl[2], dl[5], kv);
    DecoderPtrs P{dl[0], dl[3], dl[6], dl[7], dl[8], dl[9], dl[10], dl[11]};
    const int chunks = (L + 63) / 64;
    hipLaunchKernelGGL(adapter_decoder_kernel, dim3(B * chunks), dim3(64), 0, s, down32, 128, P, kv,
                       priors ? mask : (const uint8_t*)nullptr, L, Nmem, chunks, out16);
    return hipGetLastError();
}

}  // namespace hg
