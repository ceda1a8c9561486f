// Device functions of the fused attention kernels (gfx950): one 32-key tile of the online softmax for a wave's 32 queries.
// Shared by attention_kernel (hg_attn.hip) and the fused QKV projection + attention kernel (hg_qkv_attn.hip), so that the
// two are bit-identical on the same fp16 q / k / v.
#pragma once
#include "hg_kernels.h"

namespace hg {

static constexpr int HD = 64;            // head dim
static constexpr int ROWB = HD * 2;      // bytes per K/V row in LDS
static constexpr int TILEB = 32 * ROWB;  // bytes per 32-key tile

__device__ __forceinline__ int swz_k(int row) { return (row >> 1) & 7; }          // b128 row reads
// V rows: bit 2 from (row >> 1) & 1 keeps the transposing reads conflict-free (a 32-lane group reads 64 bytes of each of 4 consecutive rows:
// rows r and r + 2 must sit in different halves of the 128 bytes); the low two bits from (row >> 2) & 3 are constant inside such a group of
// 4 rows (they only permute the 16-byte chunks inside a 64-byte half) and spread the 16 rows of a register-fed ds_write_b64 (hg_qkv_attn.hip:
// one row per lane) over all 16 slots of the bank row instead of 4
__device__ __forceinline__ int swz_v(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }   // tr_b16 reads

// One 32-key tile of the online softmax for this wave's 32 queries (lane = query qcol, key half hh), in two parts so
// that a caller can reuse its Q registers in between: tile_scores() S^T = K_tile Q^T; tile_softmax_pv()
// running max / rescale, P, O^T += V_tile^T P^T.  kb / vb: LDS bases of the tile's K and V rows.  Shared by the
// whole-sequence kernel and its one-row variant, so the two are bit-identical (a streaming kernel over an LDS ring of
// key tiles and a persistent variant were built on the same two functions and measured slower: DESIGN.md section 4).
__device__ __forceinline__ void tile_scores(const char* kb, const int (&k_off)[4], const half8 (&qf)[4], f32x16& s) {
    // ---- S^T tile: lane holds keys kt*32 + (r&3) + 8*(r>>2) + 4*hh of query q
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const half8 kf = *reinterpret_cast<const half8*>(kb + k_off[ks]);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s, 0, 0, 0);
    }
}

template <bool CAUSAL>
__device__ __forceinline__ void tile_softmax_pv(const char* vb, const int (&v_off)[2], f32x16& s, const int kt, const int qt,
                                                const int q, const int L, const int rs, const int hh, const float c,
                                                float& m, float& lsum, f32x16 (&o)[2]) {
    const bool need_mask = (kt * 32 + 32 > L) || (CAUSAL && kt == qt);   // wave-uniform
    if (need_mask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            const bool ok = key < L && (!CAUSAL || key <= q);
            s[r] = ok ? s[r] : -INFINITY;
        }
    }
    float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
    for (int r = 4; r < 16; r += 4) mx = fmaxf(mx, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
    {      // the other half of the query's keys sits in lane ^ 32: v_permlane32_swap (VALU) instead of an LDS round trip
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx),
                                                         false, false);
        mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    }
    if (__any(mx > m)) {                 // some query's running max grew: rescale (wave-uniform branch)
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * c);
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    const float mc = m * c;
    half8 pf[2];
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(fmaf(s[r], c, -mc));
        ps += e;
        pf[r >> 3][r & 7] = (half_t)e;
    }
    lsum += ps;
    // ---- O^T[d][q] += sum_key V[key][d] P[q][key]; element j of lane half hh is key 16s + 8(j>>2) + 4hh + (j&3)
    // The transposing reads are inline asm: the compiler's waitcnt pass gives the builtin no memory operand and puts
    // vmcnt(0) in front of it whenever LDS-DMA is in flight; lgkmcnt is therefore waited here.
#pragma unroll
    for (int sstep = 0; sstep < 2; ++sstep) {
        if (sstep == 1 && kt * 32 + 16 >= rs) break;      // keys beyond the staged rows (all masked): wave-uniform
        fp16x4_t vr[2][2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            // the second read (keys + 8): swz_v flips bit 1 of the chunk index there = bit 5 of the byte address
            const unsigned va = (unsigned)(size_t)(HG_LDS const char*)(vb + sstep * (16 * ROWB) + v_off[dt]);
            asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3 offset:1024"
                         : "=&v"(vr[dt][0]), "=&v"(vr[dt][1])
                         : "v"(va), "v"(va ^ 32u)
                         : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vr[0][0]), "+v"(vr[0][1]), "+v"(vr[1][0]), "+v"(vr[1][1])::"memory");
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            half8 vf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vf[e] = (half_t)vr[dt][0][e];
                vf[4 + e] = (half_t)vr[dt][1][e];
            }
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[sstep], o[dt], 0, 0, 0);
        }
    }
}

}  // namespace hg
