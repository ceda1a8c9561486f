// Fused scaled-dot-product attention for gfx950, head_dim 64, whole K/V of one (sequence, head) LDS
// resident (L <= 224: 197 ViT tokens, 77 text tokens).  Replaces the SDPA inside
// nn.MultiheadAttention (clipnet/model.py:171,181-183).  The [L,L] score matrix
// never leaves registers.
//
// One workgroup (4 waves) per (sequence, head).  K and V rows (128 B each) are DMA'd global -> LDS
// (global_load_lds, swizzled on the source address).  Each wave owns 32-query tiles and computes
//   S^T = K Q^T          (v_mfma_f32_32x32x16_f16, keys on MFMA rows, queries on lanes)
// so a lane holds one query's scores: row max / sum are register reductions plus one cross-half
// exchange.  P^T (fp16) is then directly the B operand of
//   O^T = V^T P^T
// with V^T fragments fetched by the hardware transposing read ds_read_b64_tr_b16 in the k-order the
// accumulator layout dictates.  Softmax statistics and accumulation are fp32.
#include "hg_kernels.h"

namespace hg {

static constexpr int HD = 64;            // head dim
static constexpr int ROWB = HD * 2;      // bytes per K/V row in LDS

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }          // b128 row reads
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) << 2; }   // tr_b16 reads

template <int NKT, bool CAUSAL>
__global__ __launch_bounds__(256) void attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                        int L, int heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + NKT * 32 * ROWB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = heads * HD;
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const size_t ld = (size_t)3 * D;
    const half_t* base = qkv + (size_t)seq * L * ld + head * HD;

    // ---- stage K and V: piece = 8 rows x 128 B; lane -> (row = l>>3, chunk' = l&7)
    for (int piece = wave; piece < NKT * 4; piece += 4) {
        const int row = piece * 8 + (lane >> 3);
        const int src_row = row < L ? row : L - 1;
        const half_t* rp = base + (size_t)src_row * ld;
        const int cp = lane & 7;
        glds16(rp + D + ((cp ^ swz_k(row)) << 3), Ks + piece * 1024);
        glds16(rp + 2 * D + ((cp ^ swz_v(row)) << 3), Vs + piece * 1024);
    }

    __syncthreads();   // K/V landed (the barrier's fence waits for the LDS-DMA: vmcnt(0))

    const int qcol = lane & 31, hh = lane >> 5;
    const float sl2 = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 * log2(e)

    for (int qt = wave; qt * 32 < L; qt += 4) {
        // Q fragments straight from global: B operand, lane = query, k = d
        const int q = qt * 32 + qcol;
        const half_t* qp = base + (size_t)(q < L ? q : L - 1) * ld + hh * 8;
        half8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const half8*>(qp + ks * 16);

        const int nkt = CAUSAL ? (qt + 1 < NKT ? qt + 1 : NKT) : NKT;
        f32x16 s[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
            if (kt < nkt) {
                const int krow = kt * 32 + qcol;   // this lane's K row for the A fragment
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c = (2 * ks + hh) ^ swz_k(krow);
                    const half8 kf = *reinterpret_cast<const half8*>(Ks + krow * ROWB + (c << 4));
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kt], 0, 0, 0);
                }
            }
        }
        // ---- softmax over keys: lane holds keys kt*32 + (r&3) + 8*(r>>2) + 4*hh of query q
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const bool ok = (kt < nkt) && key < L && (!CAUSAL || key <= q);
                s[kt][r] = ok ? s[kt][r] : -INFINITY;
                mx = fmaxf(mx, s[kt][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
        half8 pf[NKT][2];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = exp2f((s[kt][r] - mx) * sl2);
                sum += e;
                pf[kt][r >> 3][r & 7] = (half_t)e;
            }
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---- O^T[d][q] = sum_key V[key][d] P[q][key]
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
        const int gi = lane >> 4, l16 = lane & 15;
        const int vq = l16 >> 2, vp = l16 & 3;   // tr-read address role: row vq, columns 4*vp..4*vp+3
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int sstep = 0; sstep < 2; ++sstep) {
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        // element j of lane half hh must be key 16s + 8(j>>2) + 4hh + (j&3)
                        const int key0 = kt * 32 + 16 * sstep + 4 * (gi >> 1) + vq;
                        const int chunk = dt * 4 + (gi & 1) * 2 + (vp >> 1);
                        const int a0 = key0 * ROWB + ((chunk ^ swz_v(key0)) << 4) + (vp & 1) * 8;
                        const int key1 = key0 + 8;
                        const int a1 = key1 * ROWB + ((chunk ^ swz_v(key1)) << 4) + (vp & 1) * 8;
                        const fp16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((HG_LDS fp16x4_t*)(Vs + a0));
                        const fp16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((HG_LDS fp16x4_t*)(Vs + a1));
                        half8 vf;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vf[e] = (half_t)v0[e];
                            vf[4 + e] = (half_t)v1[e];
                        }
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kt][sstep], o[dt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- store: lane = query q, d = dt*32 + (r&3) + 8*(r>>2) + 4*hh
        if (q < L) {
            half_t* op = out + ((size_t)seq * L + q) * D + head * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[dt][g * 4 + e] * inv);
                    *reinterpret_cast<half4*>(op + dt * 32 + 8 * g + 4 * hh) = h;
                }
        }
    }
}

template <int NKT, bool CAUSAL>
static hipError_t launch_t(const half_t* qkv, half_t* out, int n_seq, int L, int heads, hipStream_t s) {
    const int lds = 2 * NKT * 32 * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<NKT, CAUSAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((attention_kernel<NKT, CAUSAL>), dim3(n_seq * heads), dim3(256), lds, s, qkv, out, L, heads);
    return hipGetLastError();
}

hipError_t launch_attention(const half_t* qkv, half_t* out, int n_seq, int L, int heads, bool causal,
                            hipStream_t s) {
    if (n_seq <= 0) return hipSuccess;
    if (L < 1 || L > 224) return hipErrorInvalidValue;
    const int nkt = (L + 31) / 32;
#define HG_ATT(N)                                                                   \
    return causal ? launch_t<N, true>(qkv, out, n_seq, L, heads, s)                  \
                  : launch_t<N, false>(qkv, out, n_seq, L, heads, s)
    switch (nkt) {
        case 1: HG_ATT(1);
        case 2: HG_ATT(2);
        case 3: HG_ATT(3);
        case 4: HG_ATT(4);
        case 5: HG_ATT(5);
        case 6: HG_ATT(6);
        default: HG_ATT(7);
    }
#undef HG_ATT
}

}  // namespace hg
