// fp16 MFMA GEMM for gfx950:  D[m][n] = sum_k A[m][k] * W[n][k]   (both operands K-contiguous,
// i.e. activations [M,K] x nn.Linear weight [N,K]), fp32 accumulate, fused epilogues.
//
// Covers K3/K5/K6/K7/K8/K10 of SURVEY.md §2.2 (clipnet/model.py:171-188 linears, main_coop_vae.py
// :261-296 linears) = 95 % of the hot path's FLOPs.
//
// Structure (v1): 128x128x64 tile, 256 threads = 4 waves (2 along M x 2 along N), each wave a 64x64
// sub-tile as 4x4 v_mfma_f32_16x16x32_f16.  Operands are staged global -> LDS with 16-byte
// global_load_lds (no VGPR round trip) into two LDS buffers; the LDS image is XOR-swizzled on the
// SOURCE address (the DMA destination is lane-linear) so the ds_read_b128 fragment reads are
// bank-conflict free.  The MFMA is issued "swapped" (W rows as the A operand, activation rows as the
// B operand) so each lane ends up with 4 consecutive output columns of one output row: 8-byte fp16 /
// 16-byte fp32 epilogue accesses.
#include <stdlib.h>

#include "hg_kernels.h"

namespace hg {

static constexpr int BM = 128, BN = 128, BK = 64;
static constexpr int TILE_BYTES = 128 * BK * 2;          // 16 KiB per operand tile
static constexpr int STAGE_BYTES = 2 * TILE_BYTES;       // A + W
static constexpr int GEMM_LDS = 2 * STAGE_BYTES;         // double buffered: 64 KiB (two workgroups per CU)
static constexpr int GEMM_LDS_DEEP = 4 * STAGE_BYTES;    // four stages: 128 KiB, for grids of at most one workgroup per CU

__device__ __forceinline__ float quick_gelu(float v) {
    // x * sigmoid(1.702 x)
    float r = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -2.4554669595930157f));   // as hg_gemm_dev.h
    asm volatile("" : "+v"(r));
    return r;
}

template <int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M) return;
    if constexpr (EPI == EPI_MU_BIAS_RELU_F32) {      // W (x16 + mu) + b = acc + mu * cs + b
        const float mu = p.mu[m];
        const f32x4 cs = *reinterpret_cast<const f32x4*>(p.cs + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(mu, cs[r], v[r]);
    }
    if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v += b;
    }
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16) {
        if constexpr (EPI == EPI_BIAS_QGELU_F16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu(v[r]);
        }
        if constexpr (EPI == EPI_BIAS_RELU_F16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n) = h;
    } else if constexpr (EPI == EPI_BIAS_RESID_F32) {
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n);
        *dst = *dst + v;
    } else if constexpr (EPI == EPI_SCALE_RESID_F32) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.pos + n);
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n);
        *dst = *dst + v * sc;
    } else if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RELU_F32 || EPI == EPI_MU_BIAS_RELU_F32) {
        if constexpr (EPI == EPI_BIAS_RELU_F32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
        if constexpr (EPI == EPI_MU_BIAS_RELU_F32) {      // n_split > 0 without out_hi: columns >= n_split stay linear
            if (!(p.n_split && n >= p.n_split)) {         // (the adapter's cross-term columns ride in the padded half)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
            }
        }
        float* dst = (p.out_hi && n >= p.n_split) ? reinterpret_cast<float*>(p.out_hi) + (size_t)m * p.ldc + (n - p.n_split)
                                                  : reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n;
        *reinterpret_cast<f32x4*>(dst) = v;
    } else if constexpr (EPI == EPI_PATCH_F32) {
        const int b = m / p.G, t = m - b * p.G;
        if (p.pos) v += *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + t) * p.N + n);
        const size_t orow = (size_t)b * p.L + 1 + t;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + orow * p.ldc + n) = v;
    }
}

// STAGES = 2: tile kt+1 is fetched while tile kt is computed (64 KiB, two workgroups per CU: large grids).
// STAGES = 4: three tiles in flight behind a counted vmcnt - a small grid (M = 256 rows of the class-token stream, batch
// 1..16) is bound by the latency of its serial K loop, ~1.1 us per K-tile with one tile in flight.  Same K order:
// both variants (and the ring kernels) give the same bits.
template <int EPI, int STAGES>
__global__ __launch_bounds__(256) void gemm_nt_128x128(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    const int ntn = p.N / BN;
    const int lt = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lt / ntn, tn = lt - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- staging addresses: wave w issues DMA pieces 4w..4w+3 of each operand tile; a piece is 8
    // rows x 128 B; lane l writes LDS (row = l>>3, chunk' = l&7) and must fetch logical chunk
    // chunk' ^ swz(row), swz(row) = (row>>1)&7.
    const half_t* a_src[4];
    const half_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        int am = m0 + row;
        am = am < p.M ? am : p.M - 1;
        a_src[i] = p.A + (size_t)am * p.lda + chunk * 8;
        w_src[i] = p.W + (size_t)(n0 + row) * p.K + chunk * 8;
    }
    auto stage = [&](int buf) {
        char* base = smem + buf * STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i], base + i * 1024);
            glds16(w_src[i], base + TILE_BYTES + i * 1024);
            a_src[i] += BK;
            w_src[i] += BK;
        }
    };

    // ---- fragment read addresses (bytes inside an operand tile); swz is lane-constant because the
    // 16-row fragment bases are multiples of 16.
    const int sw = (lane >> 1) & 7;
    const int frag_row = (lane & 15) * 128;
    const int cg = lane >> 4;
    int a_off[2], w_off[2];   // per k-step; + j*16*128 per fragment
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ((ks * 4 + cg) ^ sw) << 4;
        a_off[ks] = wm * 64 * 128 + frag_row + c;
        w_off[ks] = TILE_BYTES + wn * 64 * 128 + frag_row + c;
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    constexpr int AHEAD = STAGES - 1;                    // tiles in flight
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (i < nk) stage(i);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt landed: all but the (8 DMA instructions per wave and tile) x (tiles issued after it) are done
        const int after = nk - 1 - kt < AHEAD - 1 ? nk - 1 - kt : AHEAD - 1;
        if (after >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (after == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... for every wave, and everyone is done with the buffer of tile kt - 1 (its fragments were consumed by
        // MFMAs already issued).  A raw s_barrier: __syncthreads() would add a full vmcnt(0) drain
        asm volatile("s_barrier" ::: "memory");
        if (kt + AHEAD < nk) stage((kt + AHEAD) % STAGES);
        const char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 xf[4], wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const half8*>(st + a_off[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const half8*>(st + w_off[ks] + i * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: acc[i][j][r] = D[m][n], n = n0 + wn*64 + i*16 + 4*(lane>>4) + r,
    //                                         m = m0 + wm*64 + j*16 + (lane&15)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + wm * 64 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4);
            epilogue<EPI>(p, m, n, acc[i][j]);
        }
    }
}

template <int EPI>
static hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    static bool attr_set_d[HG_MAX_DEVICES] = {};      // function attributes and CU counts are per device
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    bool& attr_set = attr_set_d[dev_i];
    int& n_cu = n_cu_d[dev_i];
    if (!attr_set) {
        n_cu = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_128x128<EPI, 2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_128x128<EPI, 4>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_DEEP);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
#ifdef HG_EXPERIMENTS
    static const bool deep_on = []() { const char* e = getenv("HG_GEMM_DEEP"); return e ? atoi(e) != 0 : true; }();
#else
    constexpr bool deep_on = true;
#endif
    const int grid = ((a.M + BM - 1) / BM) * (a.N / BN);
    if (deep_on && grid <= n_cu && a.K / BK >= 4)
        hipLaunchKernelGGL((gemm_nt_128x128<EPI, 4>), dim3(grid), dim3(256), GEMM_LDS_DEEP, s, a);
    else
        hipLaunchKernelGGL((gemm_nt_128x128<EPI, 2>), dim3(grid), dim3(256), GEMM_LDS, s, a);
    return hipGetLastError();
}

hipError_t launch_gemm(int epi, const GemmArgs& a, hipStream_t s) {
    // Which kernel runs a shape is fixed here; -DHG_EXPERIMENTS builds can override it for A/B timing: HG_GEMM=simple (the
    // 128x128 kernel everywhere), HG_DUO = 0 never / 1 plain residual GEMMs (the default) / 2 every epilogue the duo kernel
    // implements except the LayerNorm-emitting residual / 3 that one as well
#ifdef HG_EXPERIMENTS
    static const int force_simple = []() { const char* e = getenv("HG_GEMM"); return (e && e[0] == 's') ? 1 : 0; }();
    static const int duo = []() { const char* e = getenv("HG_DUO"); return e ? atoi(e) : 1; }();
#else
    constexpr int force_simple = 0, duo = 1;
#endif
    if (a.M <= 0) return hipSuccess;
    // the plain residual GEMMs (text tower, the vision tower's last c_proj) run on the two-workgroups-per-CU kernel (117 / 293 us
    // against 147 / 305 us of the ring kernels at M = 50432)
    if (epi == EPI_SCALE_RESID_LN_F32) return gemm_duo_ok(epi, a) ? launch_gemm_duo(epi, a, s) : hipErrorInvalidValue;
    if (epi == EPI_MU_BIAS_RELU_F32) return launch_gemm_simple(epi, a, s);
    const bool own_ld2 = a.ld2 && a.ld2 != a.ldc;      // fp16 copy with its own row stride: the ring kernels only
    const bool use_duo = !own_ld2 && (duo >= 3 || (duo == 2 && epi != EPI_RESID_LN_F32) || (duo == 1 && epi == EPI_BIAS_RESID_F32));
    if (!force_simple && use_duo && gemm_duo_ok(epi, a)) return launch_gemm_duo(epi, a, s);
    const bool ln = (epi == EPI_LN_BIAS_F16 || epi == EPI_LN_BIAS_QGELU_F16 || epi == EPI_RESID_LN_F32);
    if (ln) return gemm_ln_ok(epi, a) ? launch_gemm_ring(epi, a, s) : hipErrorInvalidValue;   // ring kernels only
    if (!force_simple && gemm_ring_ok(a)) return launch_gemm_ring(epi, a, s);
    return launch_gemm_simple(epi, a, s);
}

hipError_t launch_gemm_simple(int epi, const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0) return hipSuccess;
    if (a.N % BN != 0 || a.K % BK != 0 || (a.lda % 8) != 0 || (a.ldc % 4) != 0) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_BIAS_F16: return launch_t<EPI_BIAS_F16>(a, s);
        case EPI_BIAS_QGELU_F16: return launch_t<EPI_BIAS_QGELU_F16>(a, s);
        case EPI_BIAS_RELU_F16: return launch_t<EPI_BIAS_RELU_F16>(a, s);
        case EPI_BIAS_RESID_F32: return launch_t<EPI_BIAS_RESID_F32>(a, s);
        case EPI_BIAS_F32: return launch_t<EPI_BIAS_F32>(a, s);
        case EPI_PATCH_F32: return launch_t<EPI_PATCH_F32>(a, s);
        case EPI_BIAS_RELU_F32: return launch_t<EPI_BIAS_RELU_F32>(a, s);
        case EPI_SCALE_RESID_F32: return launch_t<EPI_SCALE_RESID_F32>(a, s);
        case EPI_MU_BIAS_RELU_F32: return (a.mu && a.cs) ? launch_t<EPI_MU_BIAS_RELU_F32>(a, s) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

double gemm_flops(const GemmArgs& a) { return 2.0 * a.M * (double)a.N * a.K; }

}  // namespace hg
