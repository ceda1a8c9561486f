// Fused scaled-dot-product attention for gfx950, head_dim 64, whole K/V of one (sequence, head) LDS
// resident (L <= 224: 197 ViT tokens, 77 text tokens).  Replaces the SDPA inside
// nn.MultiheadAttention (clipnet/model.py:171,181-183).  The [L,L] score matrix
// never leaves registers.
//
// One workgroup per (sequence, head) with ONE WAVE PER 32-QUERY TILE (7 waves for L = 197, 3 for
// L = 77).  K and V rows (128 B each) are DMA'd global -> LDS (global_load_lds, XOR-swizzled on the
// source address).  Each wave walks the key tiles with an online softmax:
//   S^T = K_tile Q^T      v_mfma_f32_32x32x16_f16, keys on MFMA rows, queries on lanes -> a lane holds
//                         one query's 16 scores of the tile: max / sum are register reductions plus
//                         one cross-half exchange
//   O^T += V_tile^T P^T   P^T (fp16) is directly the B operand; V^T fragments come from the hardware
//                         transposing read ds_read_b64_tr_b16 in the k-order the accumulator layout
//                         dictates
// The running max only triggers a rescale of O when some query's max actually grew (wave-uniform
// branch).  Softmax statistics and accumulation are fp32; masking (keys >= L, causal diagonal) is
// applied only on the tiles that need it.
#include <stdlib.h>

#include "hg_attn_dev.h"

namespace hg {

// ROW0: only ONE query of every sequence is wanted - row sel[seq] (row 0 when sel is null): the class token in the
// last block of the vision tower, the EOT token in the last block of the text tower; the block's other rows never
// reach the output.  Every wave helps to stage K and V, wave 0 then runs that query (all 32 lanes of the tile alias
// it) through the same instruction sequence as the full kernel, so the row is bit-identical to the full kernel's; the
// query comes from the dense matrix q0 [n_seq, D] and the result goes to a dense [n_seq, D] matrix.
// PACK (L <= 32: one query tile, one wave per item): every wave of the workgroup takes its OWN (sequence, head) item with its own
// 4 KiB (L <= 16) / 8 KiB of LDS - the generation pipeline's text tower runs 14-token prompts (4 681 x 8 items of one wave per pass
// and layer): dispatched one workgroup per item the kernel is bound by the workgroup dispatch rate (51 us), four items per workgroup
// it is not.  Same instruction sequence per wave: the item's bits do not depend on the packing.
template <bool CAUSAL, bool ROW0, bool PACK = false>
__global__ __launch_bounds__(448) void attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                        int L, int heads, int nkt, const half_t* __restrict__ q0,
                                                        const int32_t* __restrict__ sel, const int mode, const int ldo,
                                                        const int n_items) {
    // timing-experiment switches (HG_ATTN_MODE bits 1 no key loop, 2 no K/V staging, 4 no stores, 8 no Q loads; wrong results) exist
    // only in a -DHG_EXPERIMENTS build
#ifdef HG_EXPERIMENTS
    const int xmode = mode;
#else
    constexpr int xmode = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // K and V rows 0 .. rs-1 are staged, rs = L rounded up to 16 (the last key tile may be half present: its second
    // 16-key step is skipped in P V, its missing K rows read into the V region and are masked)
    const int rs = (L + 15) & ~15;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PACK: this wave is "wave 0 of 1" of its own item (waves beyond the last item redo it and store nothing)
    const int item_raw = PACK ? (int)blockIdx.x * (int)(blockDim.x >> 6) + wave_id : (int)blockIdx.x;
    const bool item_ok = !PACK || item_raw < n_items;
    const int item = item_ok ? item_raw : n_items - 1;
    const int wave = PACK ? 0 : wave_id;
    const int nwaves = PACK ? 1 : (int)(blockDim.x >> 6);
    char* const smem_item = PACK ? smem + wave_id * (2 * rs * ROWB > 4096 ? 2 * rs * ROWB : 4096) : smem;
    char* Ks = smem_item;
    char* Vs = smem_item + rs * ROWB;
    const int D = heads * HD;
    const int seq = item / heads, head = item - seq * heads;
    const size_t ld = (size_t)3 * D;
    const half_t* base = qkv + (size_t)seq * L * ld + head * HD;

    // ---- stage K and V: piece = 8 rows x 128 B; lane -> (row = l>>3, chunk' = l&7)
    for (int piece = wave; piece < ((xmode & 2) ? 0 : rs / 8); piece += nwaves) {
        const int row = piece * 8 + (lane >> 3);
        const int src_row = row < L ? row : L - 1;
        const half_t* rp = base + (size_t)src_row * ld;
        const int cp = lane & 7;
        glds16(rp + D + ((cp ^ swz_k(row)) << 3), Ks + piece * 1024);
        glds16(rp + 2 * D + ((cp ^ swz_v(row)) << 3), Vs + piece * 1024);
    }

    // ---- this wave's query tile; Q fragments straight from global (B operand: lane = query, k = d)
    int qsel = ROW0 && sel ? __builtin_amdgcn_readfirstlane(sel[seq]) : 0;
    qsel = qsel < 0 ? 0 : (qsel >= L ? L - 1 : qsel);      // caller error guard, as in layernorm_kernel
    const int qt = ROW0 ? (qsel >> 5) : wave;
    const int qcol = lane & 31, hh = lane >> 5;
    const int q = ROW0 ? qsel : qt * 32 + qcol;
    // ROW0: every lane's query aliases the sequence's row of the dense q0 matrix (only query 0 is stored)
    const half_t* qp = ROW0 ? q0 + (size_t)seq * D + head * HD + hh * 8 : base + (size_t)(q < L ? q : L - 1) * ld + hh * 8;
    half8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (xmode & 8) qf[ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};   // timing experiment: no Q loads
        else qf[ks] = *reinterpret_cast<const half8*>(qp + ks * 16);
    }

    __syncthreads();   // K/V landed (the barrier's fence waits for the LDS-DMA: vmcnt(0))
    if constexpr (ROW0) {
        if (wave != 0) return;
    }

    // lane-constant LDS offsets
    int k_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_off[ks] = qcol * ROWB + (((2 * ks + hh) ^ swz_k(qcol)) << 4);
    const int gi = lane >> 4, l16 = lane & 15;
    const int vq = l16 >> 2, vp = l16 & 3;   // tr-read role: row vq of the 4x16 block, columns 4*vp..4*vp+3
    int v_off[2];
    {
        const int key0 = 4 * (gi >> 1) + vq;   // + kt*32 + 16*sstep (+8 for the second read)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int chunk = dt * 4 + (gi & 1) * 2 + (vp >> 1);
            v_off[dt] = key0 * ROWB + ((chunk ^ swz_v(key0)) << 4) + (vp & 1) * 8;
        }
    }

    const float c = 0.125f * 1.4426950408889634f;   // head_dim^-0.5 * log2(e)
    float m = -1.0e30f, lsum = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;

    const int kt_end = (xmode & 1) ? 0 : (CAUSAL ? (qt + 1 < nkt ? qt + 1 : nkt) : nkt);
    for (int kt = 0; kt < kt_end; ++kt) {
        f32x16 sc;
        tile_scores(Ks + kt * TILEB, k_off, qf, sc);
        tile_softmax_pv<CAUSAL>(Vs + kt * TILEB, v_off, sc, kt, qt, q, L, rs, hh, c, m, lsum, o);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    const float inv = 1.0f / lsum;
    // ---- store: lane = query q, d = dt*32 + (r&3) + 8*(r>>2) + 4*hh
    if constexpr (ROW0) {
        if (qcol == 0 && !(xmode & 4)) {
            half_t* op = out + (size_t)seq * D + head * HD;
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
    } else {
        // Row-per-lane 8-byte stores touch 32 cache lines per instruction (29 us of a 100 us kernel); instead the
        // wave's 32 x 64 tile goes through LDS (the K/V rows are dead once every wave has left the key loop; 16-byte
        // chunks XOR-swizzled by row) and leaves as whole 128-byte rows: lane -> (row = l >> 3, chunk = l & 7).
        __syncthreads();
        char* ot = smem_item + wave * 4096;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[dt][g * 4 + e] * inv);
                *reinterpret_cast<half4*>(ot + qcol * 128 + (((dt * 4 + g) ^ (qcol & 7)) << 4) + hh * 8) = h;
            }
        const int cr = lane >> 3, cc = lane & 7;
#pragma unroll
        for (int rb = 0; rb < 32; rb += 8) {
            const int row = rb + cr, qq = qt * 32 + row;
            const half8 v = *reinterpret_cast<const half8*>(ot + row * 128 + ((cc ^ (row & 7)) << 4));
            if (qq < L && !(xmode & 4) && item_ok)
                *reinterpret_cast<half8*>(out + ((size_t)seq * L + qq) * ldo + head * HD + cc * 8) = v;
        }
    }
}

template <bool CAUSAL, bool ROW0 = false, bool PACK = false>
static hipError_t launch_t(const half_t* qkv, half_t* out, int n_seq, int L, int heads, hipStream_t s,
                           const half_t* q0 = nullptr, const int32_t* sel = nullptr, int ldo = 0) {
    if (ldo <= 0) ldo = heads * HD;
    const int nkt = (L + 31) / 32;
    const int lds_item = 2 * ((L + 15) & ~15) * ROWB;
    constexpr int G = 4;                              // PACK: items (waves) per workgroup
    const int lds = PACK ? G * (lds_item > 4096 ? lds_item : 4096) : lds_item;
    static bool attr_set_d[HG_MAX_DEVICES] = {};      // function attributes are per device
    bool& attr_set = attr_set_d[current_device_index()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<CAUSAL, ROW0, PACK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 7 * TILEB);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
#ifdef HG_EXPERIMENTS
    static const int mode = getenv("HG_ATTN_MODE") ? atoi(getenv("HG_ATTN_MODE")) : 0;
#else
    constexpr int mode = 0;
#endif
    const int n_items = n_seq * heads;
    if constexpr (PACK)
        hipLaunchKernelGGL((attention_kernel<CAUSAL, ROW0, true>), dim3((n_items + G - 1) / G), dim3(64 * G), lds, s, qkv, out, L, heads,
                           nkt, q0, sel, mode, ldo, n_items);
    else
        hipLaunchKernelGGL((attention_kernel<CAUSAL, ROW0, false>), dim3(n_items), dim3(64 * nkt), lds, s, qkv, out, L, heads, nkt, q0,
                           sel, mode, ldo, n_items);
    return hipGetLastError();
}

hipError_t launch_attention(const half_t* qkv, half_t* out, int n_seq, int L, int heads, bool causal,
                            hipStream_t s, int ldo, bool pack) {
    if (n_seq <= 0) return hipSuccess;
    if (L < 1 || L > 224 || (ldo != 0 && (ldo < heads * HD || ldo % 8))) return hipErrorInvalidValue;
    if (pack && L <= 32)      // one wave per item: four items per workgroup (same bits)
        return causal ? launch_t<true, false, true>(qkv, out, n_seq, L, heads, s, nullptr, nullptr, ldo)
                      : launch_t<false, false, true>(qkv, out, n_seq, L, heads, s, nullptr, nullptr, ldo);
    return causal ? launch_t<true>(qkv, out, n_seq, L, heads, s, nullptr, nullptr, ldo)
                  : launch_t<false>(qkv, out, n_seq, L, heads, s, nullptr, nullptr, ldo);
}

hipError_t launch_attention_row0(const half_t* qkv, const half_t* q0, const int32_t* sel, half_t* out, int n_seq, int L,
                                 int heads, bool causal, hipStream_t s) {
    if (n_seq <= 0) return hipSuccess;
    if (L < 1 || L > 224 || !q0) return hipErrorInvalidValue;
    return causal ? launch_t<true, true>(qkv, out, n_seq, L, heads, s, q0, sel)
                  : launch_t<false, true>(qkv, out, n_seq, L, heads, s, q0, sel);
}

}  // namespace hg
