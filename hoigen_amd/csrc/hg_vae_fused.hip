// CoOp-VAE Encoder -> reparameterise -> Generator as ONE kernel for gfx950 (main_coop_vae.py:261-296,444-448):
//
//     h  = relu(x W0^T + b0)              [rows, eh]      (Encoder.net)
//     mean | log_var = h Wm^T + bm | h Wl^T + bl          (Encoder.mean / .log_var)
//     z  = exp(0.5 log_var) * eps + mean                  (main_coop_vae.py:445-447, eps given)
//     g  = relu(z G0^T + c0)              [rows, gh]      (Generator.net.0)
//     bias = g G2^T + c2                                  (Generator.net.2)
//
// Neither hidden layer, nor z as an operand, ever reaches HBM: the only traffic is x and eps in, the four fp32 outputs out, and the
// weights streamed from L2.
//
// Structure: a two-layer perceptron the way a flash-attention kernel does S = Q K^T -> P -> O, with the DATA ROWS on the MFMA's
// column (lane) index:
//   * a wave owns 32 rows; a workgroup = 4 waves (ONE per SIMD: 512 registers each) = 128 rows = one work item;
//   * first layer, one block of 32 hidden units:  H^T[32 units][32 rows] = sum over 32 k-steps of  W0frag (A operand, from LDS) x
//     x^T frag (B operand: the wave's rows as 32 fragments of 16 k, resident in 128 registers for the whole pass);
//   * the 32 x 32 fp32 result has its row (= data row) on the lane and the 32 units in its 16 registers: + bias, relu, -> fp16 it IS
//     the B operand of the second layer's MFMAs (two k-steps of 16 units), with no lane movement and no LDS: the permutation of the 16
//     units of a k-step over (lane half, element) that the accumulator layout implies is applied to the WEIGHT fragments at pack time;
//   * second layer: out^T[512 outputs][32 rows] += W2frag (A, from LDS) x H^T frag: 16 accumulator blocks = 256 registers per wave
//     that live for the whole pass.
// Every MFMA takes exactly one 1-KiB weight fragment from LDS (ds_read_b128 at lane * 16: conflict-free, no swizzle) and all four waves
// read the same fragments: 128 B/clk/CU of the LDS's 256.  The weights are packed at load time into ONE linear stream of fragments in
// the exact order of use, so the operand fetch is a plain walk: 16-KiB stages through an 8-slot LDS ring (buffer_load ... lds, each
// wave a quarter of a stage, counted vmcnt, one s_barrier per stage five fragments before it is first read, six stages in flight).
// Bytes through the CU's load path: 16.2 MB of weights per 128 rows and 2.15 GFLOP (with the recomputation below) = 7.5 KB per MFLOP -
// the 256 x 256 GEMM tile's figure - from L2: the 32 CUs of an XCD walk the same stream in step.
//
// Passes of one item (each a walk over hidden blocks t = 0 .. nb: iteration t runs layer 1 of block t and layer 2 of block t - 1, so
// that the bias / relu / convert arithmetic of block t hides under the MFMAs of layer 2 of block t - 1):
//   E0, E1  the encoder for z columns [0, 256) and [256, 512): 8 mean + 8 log_var output blocks each (the 16 accumulator blocks hold
//           BOTH halves of the same z columns, so the reparameterisation happens on the accumulators); the encoder's hidden layer is
//           computed twice (+2.1 of 14.7 MFLOP per row) - 1024 output columns do not fit one wave's registers;
//           epilogue: + bias, z, the three fp32 outputs, and z as fp16 B fragments of the generator pass (E0's half parked in a
//           wave-private scratch, E1's half straight into the registers x leaves);
//   G       the generator on those fragments; epilogue: + bias, fp32 output.
// Generator-only calls (hg_generator) run pass G alone with z read like x.
//   M       (mode 3) the MLP of a transformer block of width 512 (the CLIP text tower: clipnet/model.py:173-177,187): the same pass with
//           QuickGELU in place of relu and the residual update as its epilogue:  x += W_proj quickgelu(W_fc h + b_fc) + b_proj,  h = the
//           fp16 LayerNorm output (row-major), x the fp32 stream in place.  c_fc's [rows, 2048] activation never reaches HBM.
//
// Arithmetic: fp16 operands (x, h, z, g, weights), fp32 accumulate and fp32 bias / relu / reparameterisation - the same roundings as the
// GEMM path of hg_api.hip (which stores h and g as fp16 in HBM), in a different summation order.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "hg_gemm_dev.h"

namespace hg {

namespace {
constexpr int VF_DIM = 512;                  // feature width (x, mean, log_var, z, bias)
constexpr int VF_KS = VF_DIM / 16;           // k-steps of a row = B fragments per wave
constexpr int VF_ROWS = 128;                 // rows per item (4 waves x 32)
#ifndef VF_STAGE_KB
#define VF_STAGE_KB 16
#endif
constexpr int VF_STAGE = VF_STAGE_KB * 1024; // one ring stage = 16 (32) fragments: one s_barrier per stage
constexpr int VF_NS = 128 / VF_STAGE_KB;     // ring slots (128 KiB)
constexpr int VF_WPIECES = VF_STAGE / 4096;  // 1-KiB pieces of a stage per wave
constexpr int VF_ITER_BYTES = 64 * 1024;     // one iteration = 64 fragments = 4 stages = half the ring
constexpr int VF_RING = 0;
constexpr int VF_TAB = VF_NS * VF_STAGE;     // first-layer bias of the pass: (nb + 2) x 32 floats
constexpr int VF_MAX_NB = 128;               // hidden <= 4096
constexpr int VF_LDS = VF_TAB + (VF_MAX_NB + 2) * 32 * 4;
#ifdef VF_AHEAD_OVERRIDE
constexpr int VF_AHEAD = VF_AHEAD_OVERRIDE;
#else
constexpr int VF_AHEAD = 4;                  // weight fragments read ahead of the MFMA that uses them
#endif
static_assert(VF_LDS <= 160 * 1024, "LDS budget");

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// iterations of a pass over nb hidden blocks: nb + 1 (layer 1 of block t beside layer 2 of block t - 1), rounded up to even
__host__ __device__ inline int vf_iters(int nb) { return (nb + 2) & ~1; }
// k index of element j of lane half h in k-step s: the order in which a 32 x 32 fp32 accumulator tile hands its 16 rows of a
// k-step to the next MFMA as a B operand (rows (r & 3) + 8 (r >> 2) + 4 h in register r)
__host__ __device__ inline int vf_kidx(int s, int h, int j) { return 16 * s + (j & 3) + 8 * (j >> 2) + 4 * h; }
}  // namespace

struct VaeFusedDev {
    const float* x;          // [R, 512] encoder input (mode 2: z)
    const half_t* x16;       // mode 2: z as fp16 [R, 512] instead (the GEMM path's reparameterisation kernel writes it)
    const float* eps;        // [R, 512]
    float* mean;             // outputs [R, 512], each nullable
    float* logvar;
    float* z;
    float* bias;
    const half_t* wp;        // packed stream of this call's passes
    const float* b0e;        // [eh]
    const float* bml;        // [1024] mean | log_var bias
    const float* b0g;        // [gh]
    const float* b2g;        // [512]
    half_t* zpark;           // [items][4 waves][16 fragments][64 lanes][8] fp16
    int R, nbe, nbg;         // hidden blocks of 32
    int mode;                // 0 Encoder + Generator, 1 Encoder, 2 Generator, 3 MLP block (QuickGELU, bias = the fp32 stream, updated in place)
    int stages_per_item;
    int n_items;
    unsigned long long* dbg;
};

// MSET: the passes this instantiation holds - 0 = Encoder (+ Generator): modes 0 and 1, 2 = Generator only, 3 = MLP block.  One kernel
// with every pass kind carried the Encoder epilogues' 36 spilled dwords (and their 68 bytes of private segment per lane) into the
// Generator-only and MLP launches, which are the ones the default dispatch uses.
template <int MSET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void vae_fused_kernel(const VaeFusedDev p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (lane-derived values are recomputed where they are used - mbcnt behind an opaque zero, so that the computation cannot be hoisted:
    // kept live across the pass loop they are what the register allocator spills, and a scratch reload waits for vmcnt(0), i.e.
    // drains the DMA ring once per iteration: measured 4 750 instead of 2 100 cycles per iteration)
    auto lane_now = [&]() {
        unsigned z = 0;
        asm volatile("" : "+s"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    };

    // timing-experiment switches (compile-time: -DVF_XMODE=bits; 1 no operand DMA, 2 every stage fetched from stream offset 0, 4 no
    // MFMA, 8 no fragment reads; wrong results) - run-time switches would put a branch at every MFMA
#ifdef VF_XMODE
    constexpr int xmode = VF_XMODE;
#else
    constexpr int xmode = 0;
#endif
#ifdef HG_STAMPS
    unsigned long long tk_vm = 0, tk_bar = 0, tk_b = 0, tk_all0 = __builtin_amdgcn_s_memtime(), tk_epi = 0, tk_x = 0;
#define VF_STAMP_B() do { tk_b = __builtin_amdgcn_s_memtime(); } while (0)
#define VF_STAMP_E(acc) do { acc += __builtin_amdgcn_s_memtime() - tk_b; } while (0)
#else
#define VF_STAMP_B() do {} while (0)
#define VF_STAMP_E(acc) do {} while (0)
#endif
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (unsigned)p.stages_per_item * VF_STAGE, 0x00020000);
    // ---- the DMA side of the weight stream: a plain walk.  d_off: stream offset of this wave's quarter of the next stage, d_lds: its
    // place in the ring; both advance by one stage per boundary and wrap (the stream at the item's end: the next item walks it again).
    // A boundary is the only non-MFMA work of the loop that does not hide under an MFMA (the wave issues in order): kept to a dozen
    // scalar instructions - one M0, the four pieces as immediate offsets (they apply to the LDS side as well).
    const int stream_bytes = p.stages_per_item * VF_STAGE;
    int d_off = wave * (VF_STAGE / 4);
    int d_lds = VF_RING + wave * (VF_STAGE / 4);
    auto issue_stage = [&]() {
        if (!(xmode & 1)) {
            const int lane16 = lane_now() * 16;
            const int off = (xmode & 2) ? wave * (VF_STAGE / 4) : d_off;
#pragma unroll
            for (int g4 = 0; g4 < VF_WPIECES / 4; ++g4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + d_lds + g4 * 4096), 16, lane16, off + g4 * 4096, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + d_lds + g4 * 4096), 16, lane16, off + g4 * 4096, 1024, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + d_lds + g4 * 4096), 16, lane16, off + g4 * 4096, 2048, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + d_lds + g4 * 4096), 16, lane16, off + g4 * 4096, 3072, 0);
            }
        }
        d_off += VF_STAGE;
        if (d_off >= stream_bytes) d_off -= stream_bytes;
        d_lds = (d_lds + VF_STAGE) & (VF_NS * VF_STAGE - 1);
    };
    // stage boundary: the stage about to be read has landed (this wave's pieces: everything but the five newer stages), every wave
    // is past the stage two back: refill its slot
    auto boundary = [&]() {
        VF_STAMP_B();
        if (!(xmode & 1)) wait_vm<(VF_NS - 3) * VF_WPIECES>();
        VF_STAMP_E(tk_vm);
        VF_STAMP_B();
        barrier_raw();
        VF_STAMP_E(tk_bar);
        issue_stage();
    };
#pragma unroll 1
    for (int i = 0; i < VF_NS - 2; ++i) issue_stage();

    half8 wr[8];             // weight fragments in flight LDS -> registers (VF_AHEAD live at a time)
    int half_sel = 0;        // which half of the ring the current iteration reads
    // (base: ring half + lane * 16, made once per iteration)
    auto rd = [&](int base, int f) { return *reinterpret_cast<const half8*>(smem + VF_RING + base + f * 1024); };
    // first boundary + read-ahead of the very first iteration
    boundary();
#pragma unroll
    for (int f = 0; f < VF_AHEAD; ++f) wr[f] = rd(lane * 16, f);

    half8 bf[VF_KS];         // B fragments of the wave's rows (x, then z)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int item = 0;

    // One pass over the hidden blocks: KIND 0 / 1 = encoder for z columns [256 KIND, 256 KIND + 256), 2 = generator.  (Instantiated
    // per kind, straight-line: a run-time kind joins the three epilogues behind one loop and the register allocator answers the
    // joins of 400 live registers with spills.)
    auto run_pass = [&](auto KIND_T, const int nb, const float* __restrict__ b0) {
        constexpr int KIND = decltype(KIND_T)::value;
        // ---- first-layer bias of the pass -> LDS (blocks nb, nb + 1: zeros - what a buffer load beyond the vector returns)
        {
            const __amdgpu_buffer_rsrc_t rsB0 = __builtin_amdgcn_make_buffer_rsrc((void*)b0, 0, (unsigned)nb * (32 * 4), 0x00020000);
            const int t_now = wave * 64 + lane_now();
            for (int i = t_now; i < (nb + 2) * 32; i += 256)
                *reinterpret_cast<unsigned*>(smem + VF_TAB + i * 4) = __builtin_amdgcn_raw_buffer_load_b32(rsB0, i * 4, 0, 0);
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            barrier_raw();
        }
        f32x16 oacc[16];
#pragma unroll
        for (int ob = 0; ob < 16; ++ob) oacc[ob] = zero16;
        // Iteration t: layer 1 of block t into hacc[C] (its first MFMA takes the bias as C operand), layer 2 of block t - 1 from hf;
        // under layer 1's MFMAs: relu + fp16 of hacc[1 - C] (block t - 1, finished an iteration ago) -> hf, one packed convert or
        // packed max per MFMA gap.  The two accumulators alternate (the loop is unrolled by two: register names are static).
        f32x16 hacc[2] = {zero16, zero16};
        half8 hf[2];
        int t = 0;
        auto iteration = [&](auto C_T) {
            constexpr int C = decltype(C_T)::value;
            const int l_now = lane_now();
            const int l16 = l_now * 16;
            const int base_cur = half_sel * VF_ITER_BYTES + l16, base_nxt = (VF_ITER_BYTES - half_sel * VF_ITER_BYTES) + l16;
            // bias of block t in this lane's accumulator order (register r: unit (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
            f32x16 bias16;
            {
                const int tab_off = VF_TAB + (t * 32 + 4 * (l_now >> 5)) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem + tab_off + 8 * g * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) bias16[4 * g + i] = b4[i];
                }
            }
            typedef _Float16 half2t __attribute__((ext_vector_type(2)));
            typedef float f32x2t __attribute__((ext_vector_type(2)));
            half2t hp2[8];
#pragma unroll
            for (int pos = 0; pos < 64; ++pos) {
                const int q = pos + VF_AHEAD;
                if ((q & (VF_STAGE / 1024 - 1)) == 0) boundary();
                if (!(xmode & 8)) wr[q & 7] = q < 64 ? rd(base_cur, q) : rd(base_nxt, q - 64);
                if (pos < 32) {
                    // layer 1 of block t: H^T += W0frag x x^T frag
                    // (inline asm: the VGPR form.  The builtin takes the AGPR form for every MFMA of the kernel, and the 256 AGPRs are
                    // the sixteen output blocks: with these accumulators there as well the allocator parks two output blocks in
                    // VGPRs and moves them in and out around each of their MFMAs - 4 x (16 writes, s_nop 11, 16 reads) per iteration)
                    if (xmode & 4) asm volatile("" ::"v"(wr[pos & 7]));
                    else if (pos == 0)
                        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(hacc[C]) : "v"(wr[pos & 7]), "v"(bf[pos]), "v"(bias16));
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(hacc[C]) : "v"(wr[pos & 7]), "v"(bf[pos]));
                    // under it: block t - 1 -> fp16.  QuickGELU (KIND 3): one element every other MFMA gap (two quarter-rate
                    // transcendentals each); relu: fp16 pairs (positions 2..9), then a packed fp16 max (10..17)
                    if constexpr (KIND == 3) {
                        if ((pos & 1) == 0) {
                            const int r = pos >> 1;
                            hf[r >> 3][r & 7] = (half_t)quick_gelu_r(hacc[1 - C][r]);
                        }
                    } else if (pos >= 2 && pos < 10) {
                        const int r = 2 * (pos - 2);
                        hp2[pos - 2] = __builtin_convertvector(f32x2t{hacc[1 - C][r], hacc[1 - C][r + 1]}, half2t);
                    } else if (pos >= 10 && pos < 18) {
                        const int k = pos - 10;
                        const half2t z2 = {0, 0};
                        const half2t m = __builtin_elementwise_max(hp2[k], z2);
                        hf[k >> 2][2 * (k & 3)] = m[0];
                        hf[k >> 2][2 * (k & 3) + 1] = m[1];
                    }
                } else {
                    // layer 2 of block t - 1: out^T[ob] += W2frag x H^T frag (k-step s2)
                    const int s2 = (pos - 32) >> 4, ob = (pos - 32) & 15;
                    if (xmode & 4) asm volatile("" ::"v"(wr[pos & 7]));
                    else oacc[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[pos & 7], hf[s2], oacc[ob], 0, 0, 0);
                }
                // (pins the order: left alone the scheduler gathers the 64 ds_reads in front of the MFMAs - 256 registers of
                // fragments in flight, the x fragments spilled)
                __builtin_amdgcn_sched_barrier(0);
            }
            half_sel ^= 1;
            ++t;
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // nb + 1 iterations, rounded up to an even count (the stream carries an all-zero iteration for it: one loop, no remainder
        // call - a second instance of the body behind a branch costs the register allocator its coalescing of 300 accumulators)
#pragma unroll 1
        for (int w = 0; w <= nb; w += 2) {
            iteration(I0{});
            iteration(I1{});
        }
        VF_STAMP_B();
        // ---- epilogue of the pass (the ring keeps filling; the next iteration's first fragments are in wr).  Buffer instructions: the
        // descriptors' sizes make rows beyond R read as zero and drop their stores, a tensor the caller did not ask for has size 0;
        // per lane ONE 32-bit offset serves every tensor (64-bit pointers per tensor were what the allocator spilled here).
        {
            const int l_now = lane_now();
            const unsigned rbytes = (unsigned)p.R * (VF_DIM * 4);
            const unsigned voff = (unsigned)(item * VF_ROWS + wave * 32 + (l_now & 31)) * (VF_DIM * 4) + 16 * (l_now >> 5);
            const unsigned boff = 16 * (l_now >> 5);
            if constexpr (KIND >= 2) {
                const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, rbytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.b2g, 0, VF_DIM * 4, 0x00020000);
#pragma unroll
                for (int ob = 0; ob < 16; ++ob) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cb = (32 * ob + 8 * g) * 4;
                        const f32x4 b2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, boff + cb, 0, 0));
                        f32x4 v = f32x4{oacc[ob][4 * g], oacc[ob][4 * g + 1], oacc[ob][4 * g + 2], oacc[ob][4 * g + 3]} + b2;
                        if constexpr (KIND == 3)      // the residual update: x += (acc + bias), this lane's own four columns in place
                            v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsO, voff + cb, 0, 0)) + v;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsO, voff + cb, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void*)p.eps, 0, rbytes, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bml, 0, 2 * VF_DIM * 4, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc((void*)p.mean, 0, p.mean ? rbytes : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)p.logvar, 0, p.logvar ? rbytes : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)p.z, 0, p.z ? rbytes : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsP =
                    __builtin_amdgcn_make_buffer_rsrc((void*)p.zpark, 0, (unsigned)p.n_items * (4 * 16 * 1024), 0x00020000);
                const unsigned poff = (unsigned)l_now * 16;
                const int psoff = (item * 4 + wave) * (16 * 1024);      // this wave's 16 parked fragments of the item
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    half8 zf[2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int cb = (256 * KIND + 32 * b + 8 * g) * 4;
                        const f32x4 bm = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, boff + cb, 0, 0));
                        const f32x4 bl = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, boff + cb + VF_DIM * 4, 0, 0));
                        const f32x4 e = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsE, voff + cb, 0, 0));
                        const f32x4 m4 = f32x4{oacc[b][4 * g], oacc[b][4 * g + 1], oacc[b][4 * g + 2], oacc[b][4 * g + 3]} + bm;
                        const f32x4 l4 = f32x4{oacc[8 + b][4 * g], oacc[8 + b][4 * g + 1], oacc[8 + b][4 * g + 2], oacc[8 + b][4 * g + 3]} + bl;
                        f32x4 z4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) z4[i] = reparam1(m4[i], l4[i], e[i]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m4), rsM, voff + cb, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, l4), rsL, voff + cb, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, z4), rsZ, voff + cb, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; ++i) zf[g >> 1][4 * (g & 1) + i] = (half_t)z4[i];
                        __builtin_amdgcn_sched_barrier(0);      // (four columns at a time: the exponentials' temporaries do not fit beside 400 live registers)
                    }
                    // z columns 256 KIND + 32 b ..: B fragments s = 16 KIND + 2 b + (0, 1) of the generator pass
                    if constexpr (KIND == 0) {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zf[0]), rsP, poff + (2 * b) * 1024, psoff, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, zf[1]), rsP, poff + (2 * b + 1) * 1024, psoff, 0);
                    } else {
                        bf[16 + 2 * b] = zf[0];
                        bf[16 + 2 * b + 1] = zf[1];
                    }
                }
                if constexpr (KIND == 1) {
                    wait_vm<0>();      // (this lane's own stores of pass E0)
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        bf[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rsP, poff + s * 1024, psoff, 0));
                }
            }
        }
        // no store of this wave is outstanding when the counted waits of the ring resume
        wait_vm<0>();
        // the first fragments of the next pass, read again (they have landed: the last boundary above was theirs): the copies the loop
        // read ahead are dead across the epilogue - twenty registers it needs
        {
            const int base = half_sel * VF_ITER_BYTES + lane_now() * 16;
#pragma unroll
            for (int f = 0; f < VF_AHEAD; ++f) wr[f] = rd(base, f);
        }
        VF_STAMP_E(tk_epi);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    for (item = blockIdx.x; item < p.n_items; item += gridDim.x) {
        // ---- the wave's rows as B fragments: element j of lane (n, h) in k-step s = x[row n][vf_kidx(s, h, j)]  (rows beyond R: zeros)
        VF_STAMP_B();
        {
            const int l_now = lane_now();
            const unsigned r_now = (unsigned)(item * VF_ROWS + wave * 32 + (l_now & 31));
            if (p.x16) {
                const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x16, 0, (unsigned)p.R * (VF_DIM * 2), 0x00020000);
                const unsigned voff = r_now * (VF_DIM * 2) + 8 * (l_now >> 5);
#pragma unroll
                for (int s = 0; s < VF_KS; ++s) {
                    const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(rsX, voff + 32 * s, 0, 0);
                    const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(rsX, voff + 32 * s + 16, 0, 0);
                    bf[s] = __builtin_bit_cast(half8, u32x4{a[0], a[1], b[0], b[1]});
                }
            } else {
                const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (unsigned)p.R * (VF_DIM * 4), 0x00020000);
                const unsigned voff = r_now * (VF_DIM * 4) + 16 * (l_now >> 5);
#pragma unroll
                for (int s8 = 0; s8 < VF_KS; s8 += 8) {
#pragma unroll
                    for (int s = s8; s < s8 + 8; ++s) {
                        const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voff + 64 * s, 0, 0));
                        const f32x4 b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, voff + 64 * s + 32, 0, 0));
                        bf[s] = half8{(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3], (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (sixteen loads in flight at a time)
                }
            }
        }
        VF_STAMP_E(tk_x);
        if constexpr (MSET == 3) run_pass(K3{}, p.nbg, p.b0g);
        else if constexpr (MSET == 2) run_pass(K2{}, p.nbg, p.b0g);
        else {
            run_pass(K0{}, p.nbe, p.b0e);
            run_pass(K1{}, p.nbe, p.b0e);
            if (p.mode != 1) run_pass(K2{}, p.nbg, p.b0g);
        }
    }
    wait_vm<0>();
#ifdef HG_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + (size_t)(blockIdx.x * 4 + wave) * 8;
        d[0] = __builtin_amdgcn_s_memtime() - tk_all0;
        d[1] = tk_vm; d[2] = tk_bar; d[3] = tk_epi; d[4] = tk_x;
    }
#endif
#endif
}

// ---- weight packing (load time).  Stream of one pass: iterations t = 0 .. nb of 64 fragments of 1 KiB:
//   f < 32:  layer-1 fragment of hidden block t, k-step f (zeros for t = nb);   f >= 32: layer-2 fragment of block t - 1, k-step
//   s2 = (f - 32) / 16, output block ob = (f - 32) % 16 (zeros for t = 0).
// Fragment = A operand of v_mfma_f32_32x32x16_f16: lane (m = l & 31, h = l >> 5), element j = W[row m][vf_kidx(s, h, j)].
// Output rows of pass kind 0 / 1: ob < 8 mean rows 256 kind + 32 ob + m, ob >= 8 log_var rows (512 +) 256 kind + 32 (ob - 8) + m;
// kind 2: rows 32 ob + m.
__global__ __launch_bounds__(256) void pack_vae_kernel(const half_t* __restrict__ W1, const half_t* __restrict__ W2, int nb, int kind,
                                                       half_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-byte piece: (t, f, lane)
    const size_t total = (size_t)vf_iters(nb) * 64 * 64;
    if (i >= total) return;
    const int lane = (int)(i & 63), f = (int)((i >> 6) & 63), t = (int)(i >> 12);
    const int m = lane & 31, h = lane >> 5;
    const int hid = nb * 32;
    half8 v = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (f < 32) {
        if (t < nb) {
            const half_t* src = W1 + (size_t)(32 * t + m) * VF_DIM;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[vf_kidx(f, h, j)];
        }
    } else if (t >= 1 && t <= nb) {
        const int s2 = (f - 32) >> 4, ob = (f - 32) & 15;
        int rowo;
        if (kind == 2) rowo = 32 * ob + m;
        else rowo = (ob < 8 ? 0 : VF_DIM) + 256 * kind + 32 * (ob & 7) + m;
        const half_t* src = W2 + (size_t)rowo * hid;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[vf_kidx(2 * (t - 1) + s2, h, j)];
    }
    *reinterpret_cast<half8*>(out + i * 8) = v;
}

size_t vae_fused_pass_bytes(int hidden) { return (size_t)vf_iters(hidden / 32) * VF_ITER_BYTES; }

bool vae_fused_ok(int dim, int eh, int gh) {
    if (dim != VF_DIM) return false;
    if (eh && (eh % 32 || eh / 32 > VF_MAX_NB)) return false;
    if (gh && (gh % 32 || gh / 32 > VF_MAX_NB)) return false;
    return true;
}

// wp: [E0 | E1 | G] (encoder and generator given), [E0 | E1] or [G]
hipError_t launch_pack_vae(const half_t* e_w0, const half_t* e_wml, int eh, const half_t* g_w0, const half_t* g_w2, int gh, half_t* wp,
                           hipStream_t s) {
    size_t off = 0;
    if (e_w0) {
        for (int kind = 0; kind < 2; ++kind) {
            const size_t total = (size_t)vf_iters(eh / 32) * 64 * 64;
            hipLaunchKernelGGL(pack_vae_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, e_w0, e_wml, eh / 32, kind,
                               wp + off / 2);
            off += vae_fused_pass_bytes(eh);
        }
    }
    if (g_w0) {
        const size_t total = (size_t)vf_iters(gh / 32) * 64 * 64;
        hipLaunchKernelGGL(pack_vae_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g_w0, g_w2, gh / 32, 2, wp + off / 2);
    }
    return hipGetLastError();
}

static hipError_t launch_vae_fused_slice(const VaeFusedArgs& a, hipStream_t s);

// The kernel addresses the [R, 512] fp32 tensors with 32-bit byte offsets: at most 2^20 rows per launch.  Longer calls run as slices of
// 2^20 rows (whole work items, whole rounds of items over 256 CUs); rows are independent, so the slicing does not change a bit.
hipError_t launch_vae_fused(const VaeFusedArgs& a, hipStream_t s) {
    constexpr int SLICE = 1 << 20;
    if (a.R <= SLICE) return launch_vae_fused_slice(a, s);
    for (int r0 = 0; r0 < a.R; r0 += SLICE) {
        VaeFusedArgs b = a;
        const size_t o = (size_t)r0 * 512;
        b.R = a.R - r0 < SLICE ? a.R - r0 : SLICE;
        if (a.x) b.x = a.x + o;
        if (a.x16) b.x16 = a.x16 + o;
        if (a.eps) b.eps = a.eps + o;
        if (a.mean) b.mean = a.mean + o;
        if (a.logvar) b.logvar = a.logvar + o;
        if (a.z) b.z = a.z + o;
        if (a.bias) b.bias = a.bias + o;
        if (a.zpark) b.zpark = a.zpark + vae_fused_park_bytes(r0) / sizeof(half_t);
        hipError_t e = launch_vae_fused_slice(b, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

static hipError_t launch_vae_fused_slice(const VaeFusedArgs& a, hipStream_t s) {
    if (a.R <= 0 || (!a.x && !(a.mode >= 2 && a.x16)) || !a.wp || a.mode < 0 || a.mode > 3) return hipErrorInvalidValue;
    if (a.mode < 2 && (!a.eps || !a.b0e || !a.bml || !a.zpark)) return hipErrorInvalidValue;
    if (a.mode != 1 && (!a.b0g || !a.b2g || !a.bias)) return hipErrorInvalidValue;
    VaeFusedDev d{};
    d.x = a.x; d.x16 = a.mode >= 2 ? a.x16 : nullptr; d.eps = a.eps; d.mean = a.mean; d.logvar = a.logvar; d.z = a.z; d.bias = a.bias;
    d.b0e = a.b0e; d.bml = a.bml; d.b0g = a.b0g; d.b2g = a.b2g; d.zpark = a.zpark;
    d.R = a.R; d.nbe = a.eh / 32; d.nbg = a.gh / 32;
    const size_t enc_bytes = 2 * vae_fused_pass_bytes(a.eh), gen_bytes = vae_fused_pass_bytes(a.gh);
    size_t bytes = 0;
    d.wp = a.wp;
    d.mode = a.mode;
    if (a.mode == 0) bytes = enc_bytes + gen_bytes;
    else if (a.mode == 1) bytes = enc_bytes;
    else if (a.mode == 2) { d.wp = a.wp + (a.has_enc ? enc_bytes / 2 : 0); bytes = gen_bytes; }
    else bytes = gen_bytes;      // mode 3: wp is the one pass of this block's MLP
    if (bytes >= (1ull << 31) || a.R > (1 << 20)) return hipErrorInvalidValue;      // (32-bit byte offsets into the [R, 512] fp32 tensors)
    d.stages_per_item = (int)(bytes / VF_STAGE);
    d.n_items = (a.R + VF_ROWS - 1) / VF_ROWS;
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    if (!attr_set_d[dev_i]) {
        n_cu_d[dev_i] = 256;
        for (const void* f : {reinterpret_cast<const void*>(&vae_fused_kernel<0>), reinterpret_cast<const void*>(&vae_fused_kernel<2>),
                              reinterpret_cast<const void*>(&vae_fused_kernel<3>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu_d[dev_i] = prop.multiProcessorCount;
        attr_set_d[dev_i] = true;
    }
    const int grid = d.n_items < n_cu_d[dev_i] ? d.n_items : n_cu_d[dev_i];
    d.dbg = nullptr;
#ifdef HG_STAMPS
    static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc((void**)&dbg, (size_t)1024 * 4 * 8 * 8);
    d.dbg = dbg;
    if (dbg) (void)hipMemsetAsync(dbg, 0, (size_t)1024 * 4 * 8 * 8, s);
#endif
    if (a.mode == 3) hipLaunchKernelGGL(vae_fused_kernel<3>, dim3(grid), dim3(256), VF_LDS, s, d);
    else if (a.mode == 2) hipLaunchKernelGGL(vae_fused_kernel<2>, dim3(grid), dim3(256), VF_LDS, s, d);
    else hipLaunchKernelGGL(vae_fused_kernel<0>, dim3(grid), dim3(256), VF_LDS, s, d);
#ifdef HG_STAMPS
    if (dbg && getenv("HG_VF_STAMPS")) {
        (void)hipStreamSynchronize(s);
        static std::vector<unsigned long long> h;
        h.resize((size_t)grid * 4 * 8);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const char* nm[5] = {"total", "vmcnt wait", "barrier", "epilogues", "x load"};
        fprintf(stderr, "vae_fused stamps (100 MHz ticks, median over %d waves; rows %d mode %d):", grid * 4, a.R, a.mode);
        for (int k = 0; k < 5; ++k) {
            std::vector<unsigned long long> v;
            for (int w = 0; w < grid * 4; ++w) v.push_back(h[(size_t)w * 8 + k]);
            std::sort(v.begin(), v.end());
            fprintf(stderr, " %s %llu (max %llu) |", nm[k], v[v.size() / 2], v.back());
        }
        fprintf(stderr, "\n");
    }
#endif
    return hipGetLastError();
}

int vae_fused_rows_per_item() { return VF_ROWS; }

}  // namespace hg
