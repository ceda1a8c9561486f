// Device helpers shared by the persistent ring GEMM kernels (hg_gemm_ring.hip, hg_gemm_ring2.hip).
#pragma once
#include "hg_kernels.h"

namespace hg {

// x * sigmoid(1.702 x) = x / (1 + 2^(-1.702 log2(e) x)): one constant multiply and the hardware exp2 / rcp, written
// out so that every kernel (and every code layout) rounds identically (a row's result must not depend on which
// kernel its batch size selects) - __expf leaves the compiler free to merge or
// not merge its log2(e) multiply with ours
__device__ __forceinline__ float quick_gelu_r(float v) {
    float r = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -2.4554669595930157f));
    // keep the product a rounded fp32 value: otherwise the compiler fuses this multiply with the fp16 conversion
    // (v_fma_mixlo_f16, one rounding) for some elements and not for others, differently in every kernel
    asm volatile("" : "+v"(r));
    return r;
}

// Four elements at once: the two multiplies and the add are written on vectors so that they become packed fp32
// instructions (v_pk_mul_f32 / v_pk_add_f32: two elements per issue; the epilogue is VALU-bound, two thirds of it the
// quarter-rate exp and rcp).  Same IEEE operations per element as quick_gelu_r: bit-identical results.
// `kk` = {k, k, 1, 1} with k = -1.702 * log2(e), made opaque once per epilogue by quick_gelu_consts(): with literal
// constants the compiler picks the scalar v_mul_f32 / v_add_f32 (a packed instruction cannot take a literal).
__device__ __forceinline__ f32x4 quick_gelu_consts() {
    f32x4 kk = {-2.4554669595930157f, -2.4554669595930157f, 1.0f, 1.0f};
    asm volatile("" : "+v"(kk));
    return kk;
}
__device__ __forceinline__ f32x4 quick_gelu4(f32x4 v, f32x4 kk) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 k2 = {kk[0], kk[1]}, one2 = {kk[2], kk[3]};
    f32x4 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 x = {v[2 * h], v[2 * h + 1]};
        const f32x2 t = x * k2;
        f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        const f32x2 d = e + one2;
        const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        f32x2 p = x * r;
        asm volatile("" : "+v"(p));      // keep the products rounded fp32 values (see quick_gelu_r)
        o[2 * h] = p[0];
        o[2 * h + 1] = p[1];
    }
    return o;
}

// z = exp(0.5 * log_var) * eps + mean (main_coop_vae.py:445-447), one expression for the stand-alone kernel and the GEMM
// epilogue (a row's z must not depend on which of the two its chunk size selects)
__device__ __forceinline__ float reparam1(float mean, float log_var, float eps) {
    return __builtin_fmaf(expf(0.5f * log_var), eps, mean);
}

template <int EPI>
__device__ __forceinline__ void epilogue_ring(const GemmArgs& p, int m, int n, f32x4 v) {
    if (m >= p.M) return;
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_QGELU_F16 || EPI == EPI_BIAS_RELU_F16) {
        if constexpr (EPI == EPI_BIAS_QGELU_F16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = quick_gelu_r(v[r]);
        }
        if constexpr (EPI == EPI_BIAS_RELU_F16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (half_t)v[r];
        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.out) + (size_t)m * p.ldc + n) = h;
    } else if constexpr (EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_RELU_F32) {
        if constexpr (EPI == EPI_BIAS_RELU_F32) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        }
        float* dst = (p.out_hi && n >= p.n_split) ? reinterpret_cast<float*>(p.out_hi) + (size_t)m * p.ldc + (n - p.n_split)
                                                  : reinterpret_cast<float*>(p.out) + (size_t)m * p.ldc + n;
        *reinterpret_cast<f32x4*>(dst) = v;
    } else if constexpr (EPI == EPI_PATCH_F32) {
        const int b = m / p.G, t = m - b * p.G;
        // pos == nullptr (the product path): the positional embedding is added by the ln_pre kernel, which reads the rows
        // anyway - a register-returning global load here waits for every older DMA of the ring
        if (p.pos) v += *reinterpret_cast<const f32x4*>(p.pos + (size_t)(1 + t) * p.N + n);
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + ((size_t)b * p.L + 1 + t) * p.ldc + n) = v;
    }
}

// x + x(lane ^ 16) + x(lane ^ 32) + x(lane ^ 48): the sum over the four 16-lane rows, in every lane, with two
// v_permlane*_swap instead of four ds_bpermute round trips.  v_permlane16_swap exchanges the odd rows of its first
// operand with the even rows of the second, v_permlane32_swap the upper half of the first with the lower half of
// the second; called with the same value twice, first + second is the pairwise sum.
__device__ __forceinline__ float sum_rows(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float y = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned v = __builtin_bit_cast(unsigned, y);
    const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// LayerNorm statistics of one row from the partial sums the residual GEMMs emit (finalize_stats_kernel, hg_elem.hip; the tail of the
// MLP pair kernel, hg_mlp_pair.hip - one function, so that a row's (mean, rstd) do not depend on which of the two ran): stats [nt][2] =
// per column group of gw columns (sum, sum of squared deviations from the group mean), combined with Chan's formula -> mr = (mean - c, rstd)
// with c = mu[m] the centre the row's fp16 copy was written with (centred: mr = (mean, rstd), mu stays), then mu[m] = mean, muc[m] = c.
// SC1: the partial sums were written by other workgroups of this launch (same XCD): read them past this CU's vector L1.
template <bool SC1>
__device__ __forceinline__ void finalize_stats_row(const float* sp, float* mr, float* mu, float* muc, int m, int nt, int gw, int centred,
                                                   int* range_flag) {
    float s1 = 0.f, m2 = 0.f;
    const float D = (float)(nt * gw);
    float mean;
    const float c = mu[m];
    float reach = 0.f;      // upper bound of |x - c| over the row: per column group, sqrt(sum of squared deviations) + |group mean - c|
    auto ld4 = [&](int i) {
        if constexpr (SC1) {
            typedef float f32x4g __attribute__((ext_vector_type(4)));
            f32x4 v;
            const f32x4g* q = reinterpret_cast<const f32x4g*>(sp) + i;
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(q) : "memory");
            return v;
        } else {
            return reinterpret_cast<const f32x4*>(sp)[i];
        }
    };
    if (nt == 12) {      // D = 768: the row's 24 floats as six 16-byte loads (same order of additions as the loop below)
        f32x4 v[6];
        if constexpr (SC1) {      // all six in flight, ONE wait (a wait per load costs six L2 round trips: 10 us in the pair kernel's tail)
            typedef float f32x4g __attribute__((ext_vector_type(4)));
            const f32x4g* q = reinterpret_cast<const f32x4g*>(sp);
            asm volatile("global_load_dwordx4 %0, %6, off sc1\n global_load_dwordx4 %1, %6, off offset:16 sc1\n"
                         "global_load_dwordx4 %2, %6, off offset:32 sc1\n global_load_dwordx4 %3, %6, off offset:48 sc1\n"
                         "global_load_dwordx4 %4, %6, off offset:64 sc1\n global_load_dwordx4 %5, %6, off offset:80 sc1\n s_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]) : "v"(q) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) v[i] = ld4(i);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) { s1 += v[i][0]; s1 += v[i][2]; }
        mean = s1 / D;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float d0 = v[i][0] / (float)gw - mean;
            m2 += v[i][1] + (float)gw * d0 * d0;
            const float d1 = v[i][2] / (float)gw - mean;
            m2 += v[i][3] + (float)gw * d1 * d1;
            reach = fmaxf(reach, fmaxf(sqrtf(v[i][1]) + fabsf(v[i][0] / (float)gw - c), sqrtf(v[i][3]) + fabsf(v[i][2] / (float)gw - c)));
        }
    } else {
        for (int t = 0; t < nt; t += 2) {      // (nt = 4 * N / 256 is even)
            const f32x4 v = ld4(t / 2);
            s1 += v[0];
            s1 += v[2];
        }
        mean = s1 / D;
        for (int t = 0; t < nt; t += 2) {
            const f32x4 v = ld4(t / 2);
            const float d0 = v[0] / (float)gw - mean;
            m2 += v[1] + (float)gw * d0 * d0;
            reach = fmaxf(reach, sqrtf(v[1]) + fabsf(v[0] / (float)gw - c));
            const float d1 = v[2] / (float)gw - mean;
            m2 += v[3] + (float)gw * d1 * d1;
            reach = fmaxf(reach, sqrtf(v[3]) + fabsf(v[2] / (float)gw - c));
        }
    }
    mr[2 * (size_t)m] = centred ? mean : mean - c;   // the fp16 copy of this row was written as x - mu[m]
    mr[2 * (size_t)m + 1] = 1.0f / sqrtf(m2 / D + 1e-5f);
    // The copy of this row the folded GEMMs read (and, with the stream held as hi / lo, the stream itself) was just written as
    // fp16(x - c): an element more than 65 504 from the centre overflows it.  The statistics bound the row's reach from above (no
    // element is further from c than its group's root sum of squared deviations + the group mean's distance), so a row is reported
    // when that bound leaves the range - before anything has turned non-finite, and never for a row that came in non-finite (NaN
    // compares false).  The word is host-mapped and sticky: the NEXT tower call reports it (hg_api.hip).
    if (range_flag && !centred && reach > 65504.0f) *range_flag = 1;
    if (muc) muc[m] = c;                          // centre of the current copy (adapter down_proj adds it back)
    if (!centred) mu[m] = mean;                   // centre for the next residual GEMM's copy
}

// Tile schedule of the stand-alone persistent kernels (gemm_ring, gemm_ring2).  Tile order (L2 locality): the list is n-group-major
// (groups of `gsz` column tiles whose W slices fit one XCD's L2 together), m-tile next, column tile inside the group fastest.  XCD x
// (= blockIdx % 8 under round-robin placement; speed only) owns the contiguous list range [x*T8, (x+1)*T8) and its CUs walk it 'cpx'
// items per round, so an XCD keeps re-using the same W slices while streaming A panels.  Workgroup's tiles: slot, slot + cpx, ...
// The kernel bodies (hg_gemm_ring_body.h, hg_gemm_ring2_body.h) only see n_items() / tile() / slack() and the hand-off switches.
struct RingTileList {
    static constexpr bool PUBLISH = false, CONSUME = false;      // no tile is handed to / taken from another workgroup of the launch
    int slot, cpx, my_tiles, max_tiles, gsz, ngf, grem, per_grp;
    __device__ __forceinline__ RingTileList(int n_tiles, int tiles_n, int gsz_) {
        const int G = gridDim.x, bid = blockIdx.x;
        const bool xcd_ok = (G & 7) == 0;
        cpx = xcd_ok ? (G >> 3) : G;                                   // workgroups per XCD
        const int T8 = xcd_ok ? (n_tiles + 7) / 8 : n_tiles;            // list items per XCD
        const int xbase = xcd_ok ? (bid & 7) * T8 : 0;
        const int xend = (xbase + T8 < n_tiles) ? xbase + T8 : n_tiles;
        slot = xbase + (xcd_ok ? (bid >> 3) : bid);                    // first list item of this workgroup
        my_tiles = slot < xend ? (xend - slot + cpx - 1) / cpx : 0;
        max_tiles = (T8 + cpx - 1) / cpx;
        gsz = gsz_;
        const int tiles_m_all = n_tiles / tiles_n;
        ngf = tiles_n / gsz;
        grem = tiles_n - ngf * gsz;
        per_grp = tiles_m_all * gsz;
    }
    __device__ __forceinline__ int n_items() const { return my_tiles; }
    __device__ __forceinline__ int slack() const { return max_tiles - my_tiles; }      // tiles fewer than the fullest workgroups own
    __device__ __forceinline__ void tile(int r, int& tm, int& tn) const {
        const int item = slot + r * cpx;
        if (item < ngf * per_grp) {
            const int grp = item / per_grp, rr = item - grp * per_grp;
            tm = rr / gsz;
            tn = grp * gsz + (rr - tm * gsz);
        } else {
            const int rr = item - ngf * per_grp;
            tm = rr / grem;
            tn = ngf * gsz + (rr - tm * grem);
        }
    }
    __device__ __forceinline__ int thread_id() const { return threadIdx.x; }
    // (hand-off hooks of the MLP pair kernel's schedules; never called on this one)
    __device__ __forceinline__ void publish(int, int) const {}
    __device__ __forceinline__ unsigned poll_issue(int) const { return 0u; }
    __device__ __forceinline__ void poll_finish(int, unsigned) const {}
    __device__ __forceinline__ void poll_blocking(int) const {}
};

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void barrier_raw() { asm volatile("s_barrier" ::: "memory"); }

}  // namespace hg
