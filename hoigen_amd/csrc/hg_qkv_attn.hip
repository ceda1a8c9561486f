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

#define QA_KERNEL qkv_attn_kernel
#define SQ_NKMOD 0
#include "hg_qkv_attn_body.inc"
#undef QA_KERNEL
#undef SQ_NKMOD
// K = D + 64 (13 K-tiles at D = 768): variant C's in_proj over [x16 | e] with the adapter folded into it (hg_api.hip run_blocks)
#define QA_KERNEL qkv_attn_kernel_k1
#define SQ_NKMOD 1
#include "hg_qkv_attn_body.inc"
#undef QA_KERNEL
#undef SQ_NKMOD

// ---- weight packing (load time): the LayerNorm-folded in_proj weight [3D, D] into fragment order, bias' and column sums
// into tile column order.  Tile column block lb = 3 wave + c of head pair hp: head 2 hp + lb / 12, matrix (lb % 12) / 4 (q, k, v),
// columns 16 ((lb % 12) % 4) .. + 15 of that head.
__global__ __launch_bounds__(256) void pack_qkv_kernel(const half_t* __restrict__ W, const float* __restrict__ bias,
                                                       const float* __restrict__ cs, half_t* __restrict__ Wp,
                                                       float* __restrict__ bcs, int D, int heads, int K) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece: (hp, k32, wave, c, lane)
    const int HP = heads / 2, K32 = K / 32;      // (K = row length of W: D, or D + 64 with the adapter folded in)
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
        *reinterpret_cast<half8*>(Wp + i * 8) = *reinterpret_cast<const half8*>(W + (size_t)n * K + 32 * k32 + 8 * (lane >> 4));
    }
    if (bcs && i < (size_t)HP * 384) {
        const int hp = (int)(i / 384), j = (int)(i % 384);
        const int n = src_row(hp, j / 16, j % 16);
        bcs[hp * 768 + j] = bias ? bias[n] : 0.f;
        bcs[hp * 768 + 384 + j] = cs[n];
    }
}

hipError_t launch_pack_qkv(const half_t* W, const float* bias, const float* cs, half_t* Wp, float* bcs, int D, int heads,
                           hipStream_t s, int K) {
    if (K <= 0) K = D;
    if (heads < 2 || (heads & 1) || D != heads * 64 || !W || !Wp || (bcs && !cs) || K % 64) return hipErrorInvalidValue;
    const size_t total = (size_t)(heads / 2) * (K / 32) * 8 * QA_NCB * 64;
    hipLaunchKernelGGL(pack_qkv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, bias, cs, Wp, bcs, D, heads, K);
    return hipGetLastError();
}

bool qkv_attn_ok(int n_seq, int L, int D, int heads, int lda, int K) {
    if (K <= 0) K = D;
    if (n_seq < 1 || heads < 2 || (heads & 1) || D != heads * 64) return false;
    if (L <= 192 || L > QA_RB * 16) return false;                 // seven 32-key tiles, thirteen 16-row blocks
    if ((D / 64) % 3 || D / 64 < 6) return false;                 // 384-column panels = head pairs
    if (K % 64 || (K / 64) % 3 == 2 || K / 64 < 6 + (K / 64) % 3) return false;      // K-tile schedules: 3 m (>= 6) and 3 m + 1 (>= 7)
    if (lda < K || (lda & 7)) return false;
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
    if (a.K <= 0) a.K = a.D;
    if (!qkv_attn_ok(a.n_seq, a.L, a.D, a.heads, a.lda, a.K) || !a.x16 || !a.wp || !a.bcs || !a.mr || !a.out) return hipErrorInvalidValue;
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
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_attn_kernel_k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
    if ((a.K / 64) % 3 == 0) hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(512), QA_LDS, s, a, mode);
    else hipLaunchKernelGGL(qkv_attn_kernel_k1, dim3(grid), dim3(512), QA_LDS, s, a, mode);      // (K = 64 (3 m + 1): the adapter folded in)
    return hipGetLastError();
}

}  // namespace hg
