// Residual GEMM of the ViT block with a DEFERRED, WHOLE-LINE epilogue (gfx950):
//
//   x[m][n] += sum_k A[m][k] W[n][k] + bias[n];   x16 = fp16(x - mu[m]);   (sum, M2) per row and 64-column group
//
// i.e. EPI_RESID_LN_F32 (attention out-projection and MLP c_proj, clipnet/model.py:185-188, with the extras the
// LayerNorm-folded consumer GEMMs need).  Same K loop as hg_gemm_ring2.hip (128 x 256 x 64 tiles, 8 waves as 2 x 4, three
// 48 KiB LDS stages filled by buffer_load ... lds, two [fetch | 16 MFMAs] phases per K-tile, waves 4-7 one barrier interval
// behind waves 0-3).  What differs is where the 387 MB of epilogue traffic go ("Round 2"): in ring2 every CU
// reads its 128 KiB of residual rows in the last K-tile and writes 192 KiB in a burst of 28 store instructions per wave
// whose 16-row x 64-byte patterns hold the CU's store path ~72 cycles each.  Here
//
//   * every global access of the epilogue covers WHOLE 128-byte lines: the W rows are permuted at DMA time so that a wave
//     owns 64 consecutive columns and a lane's four MFMA blocks 16 consecutive ones, and the accumulators of a 16-row
//     block are transposed across the lanes {r, r+4, r+8, r+12} of a 16-lane row (two v_mov_dpp per register) into the
//     "T layout": slot j of lane (r0 + 4i, q) = row r0 + 4j, columns 16q + 4i .. +3 - a store instruction then writes
//     four rows x 256 bytes;
//   * the updated fp32 rows are PARKED in the 64 registers the residual rows came in (they are the same rows) and leave
//     four store instructions per K-tile under the next tile's K-tiles 1-4; the next tile's residual rows arrive through
//     the same registers, four loads per K-tile, under K-tiles 5-8.  Only the fp16 copy (8 stores) and the statistics
//     remain a burst at the tile boundary.
//
// Counted vmcnt: the trickle operations sit at a fixed place (PB fetch segment, before the W refill), so every wait of the
// loop knows how many younger operations are in flight; K-tiles are instantiated by position as in ring2.
// Requirements beyond ring2's: M % 128 == 0 (no ragged tile: a masked store instruction could vanish from the count),
// K >= 768 (12 K-tiles: 1 + 8 trickle K-tiles + the last three).  Arithmetic per element is ring2's, bit for bit
// (x and x16; the row statistics reduce in a different lane order).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "hg_gemm_dev.h"

namespace hg {

namespace {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// DPP controls (probed on gfx950, tools/ubench/dpp_probe.hip): row_ror:8 swaps the halves of a 16-lane row, row_shr:4 reads
// lane l - 4, row_shl:4 lane l + 4; bank_mask bit k enables lanes 4k .. 4k+3 of every row, disabled lanes keep `old`.
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, 0xF, BANK, false));
}
template <int CTRL, int BANK>
__device__ __forceinline__ unsigned dpp_u(unsigned old, unsigned src) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, BANK, false);
}

// 4 x 4 transpose between the register index b and the lane index i = (lane >> 2) & 3 of a 16-lane row (lanes r0 + 4i share
// r0 = lane & 3):  out[j] at lane i  =  in[i] of lane j.  Two butterfly stages, two v_mov_dpp per register.
__device__ __forceinline__ void xpose4(float (&v)[4]) {
    // stage 1: bit 1 of i <-> bit 1 of the register index (partner = lane ^ 8)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const float X = v[b], Y = v[b + 2];
        v[b] = dpp_f<0x128, 0xC>(X, Y);        // lanes 8-15: partner's register b + 2
        v[b + 2] = dpp_f<0x128, 0x3>(Y, X);    // lanes 0-7:  partner's register b
    }
    // stage 2: bit 0 of i <-> bit 0 of the register index (partner = lane ^ 4)
#pragma unroll
    for (int b = 0; b < 4; b += 2) {
        const float X = v[b], Y = v[b + 1];
        v[b] = dpp_f<0x114, 0xA>(X, Y);        // lanes with i odd: lane - 4's register b + 1
        v[b + 1] = dpp_f<0x104, 0x5>(Y, X);    // lanes with i even: lane + 4's register b
    }
}

// sum over the four lanes r0 + 4i (i = 0..3) of a 16-lane row, in every one of them
__device__ __forceinline__ float sum_i(float s) {
    s += dpp_f<0x128, 0xF>(s, s);
    s += dpp_f<0x124, 0xF>(s, s);
    return s;
}

}  // namespace

__global__ __launch_bounds__(512, 2) void gemm_ring2t(const GemmArgs p, const int tiles_n, const int n_tiles,
                                                      const unsigned a_bytes, const int mode, const int gsz) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BM = 128, BK = 64;
    constexpr int AB = 16384, WH = 16384;              // bytes: A tile (both halves), one W half
    constexpr int STAGE = AB + 2 * WH;                 // 48 KiB
    constexpr int NST = 3;
    constexpr int GA = 2, GW = 4;                      // DMA instructions per wave: A tile, both W halves
    constexpr int NWT = GW + GA + GW;                  // younger DMAs when A,W(t+1) must have landed
    constexpr int E = 8 + 16;                          // burst stores per wave at a tile boundary: fp16 copy, statistics
    constexpr int R = 4;                               // row-centre loads per wave in the last K-tile
    constexpr int BIAS_OFF = NST * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = p.K / BK;

    // ---- tile list (as in ring2): XCD-contiguous chunks, column tile of a row panel fastest
    const int G = gridDim.x, bid = blockIdx.x;
    const bool xcd_ok = (G & 7) == 0;
    const int cpx = xcd_ok ? (G >> 3) : G;
    const int T8 = xcd_ok ? (n_tiles + 7) / 8 : n_tiles;
    const int xbase = xcd_ok ? (bid & 7) * T8 : 0;
    const int xend = (xbase + T8 < n_tiles) ? xbase + T8 : n_tiles;
    const int slot = xbase + (xcd_ok ? (bid >> 3) : bid);
    const int my_tiles = slot < xend ? (xend - slot + cpx - 1) / cpx : 0;
    const int tiles_m_all = n_tiles / tiles_n;
    const int ngf = tiles_n / gsz, grem = tiles_n - ngf * gsz, per_grp = tiles_m_all * gsz;
    auto tile_of = [&](int item, int& tm, int& tn) {
        if (item < ngf * per_grp) {
            const int grp = item / per_grp, rr = item - grp * per_grp;
            tm = rr / gsz;
            tn = grp * gsz + (rr - tm * gsz);
        } else {
            const int rr = item - ngf * per_grp;
            tm = rr / grem;
            tn = ngf * gsz + (rr - tm * grem);
        }
    };
    if (my_tiles <= 0) return;
    {      // start stagger of the workgroups with slack (see ring2)
        const int dunit = mode >> 8;
        const int max_tiles = (T8 + cpx - 1) / cpx;
        if (dunit > 0 && my_tiles < max_tiles) {
            const unsigned h = ((unsigned)bid * 2654435761u) >> 24;
            const long long d = ((long long)(max_tiles - my_tiles) * nk * dunit * h) >> 8;
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            while ((long long)(__builtin_amdgcn_s_memtime() - t0) < d) __builtin_amdgcn_s_sleep(32);
        }
    }
    const int S = my_tiles * nk;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * 2), 0x00020000);

    // ---- DMA source offsets: a piece is 8 rows x 128 B; lane -> (row = l>>3, chunk' = l&7)
    int voffA[GA], voffW[GW];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + (lane >> 3);            // 0..127
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        voffA[i] = row * p.lda * 2 + c * 16 - i * 1024;
    }
#pragma unroll
    for (int i = 0; i < GW; ++i) {
        const int row = (wave * GW + i) * 8 + (lane >> 3);            // LDS row 0..255 (W half h = row >> 7)
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        // LDS row (h, w = rr >> 5, g2 = (rr >> 4) & 1, i16 = rr & 15) holds tile column w*64 + 16*(i16 >> 2) + 4*(2h + g2) +
        // (i16 & 3): wave w owns the 64 consecutive columns w*64.., a lane's four MFMA blocks (hb, g2) 16 consecutive ones
        const int h = row >> 7, rr = row & 127;
        const int src = (rr >> 5) * 64 + 16 * ((rr & 15) >> 2) + 4 * (2 * h + ((rr & 31) >> 4)) + (rr & 3);
        voffW[i] = src * p.K * 2 + c * 16 - i * 1024;
    }
    struct Ld { int kt, r, soff, st; };
    Ld lA{nk - 1, -1, 0, (NST - 1) * STAGE}, lW{nk - 1, -1, 0, (NST - 1) * STAGE};
    auto advance = [&](Ld& l, bool isA, auto WRAP_T) {
        constexpr int WRAP = decltype(WRAP_T)::value;
        ++l.kt;
        if (WRAP == 1 || (WRAP == 2 && l.kt == nk)) {
            l.kt = 0;
            ++l.r;
            int tm, tn;
            tile_of(slot + l.r * cpx, tm, tn);
            l.soff = isA ? tm * BM * p.lda * 2 : tn * 256 * p.K * 2;
        }
        l.st = l.st == (NST - 1) * STAGE ? 0 : l.st + STAGE;
    };
    auto dma_A = [&](auto I) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (HG_LDS void*)(smem + lA.st + wave * GA * 1024), 16, voffA[i],
                                                 lA.soff + lA.kt * (BK * 2), i * 1024, 0);
    };
    auto dma_W = [&](auto I) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (HG_LDS void*)(smem + lW.st + AB + wave * GW * 1024), 16, voffW[i],
                                                 lW.soff + lW.kt * (BK * 2), i * 1024, 0);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using P2 = std::integral_constant<int, 2>;
    using P3 = std::integral_constant<int, 3>;
    auto issue_A = [&](auto WRAP_T) {
        advance(lA, true, WRAP_T);
        dma_A(P0{}); dma_A(P1{});
    };
    auto issue_W = [&](auto WRAP_T) {
        advance(lW, false, WRAP_T);
        dma_W(P0{}); dma_W(P1{}); dma_W(P2{}); dma_W(P3{});
    };
    using WDYN = std::integral_constant<int, 2>;

    // ---- fragment read offsets
    const int sw = (lane >> 1) & 7;
    int coff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) coff[ks] = ((ks * 4 + (lane >> 4)) ^ sw) << 4;
    const int a_row = (wm * 32 + (lane & 15)) * 128;                 // + ha*8192 + f*2048
    const int w_row = AB + (wn * 32 + (lane & 15)) * 128;            // + hb*WH + g2*2048

    half8 xa[2][2], wb[2][2][2];
    auto read_A = [&](int ha, int st) {
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xa[f][ks] = *reinterpret_cast<const half8*>(smem + st + ha * 8192 + a_row + f * 2048 + coff[ks]);
    };
    auto read_W = [&](int st) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    wb[hb][g2][ks] = *reinterpret_cast<const half8*>(smem + st + hb * WH + w_row + g2 * 2048 + coff[ks]);
    };
    f32x4 acc[2][2][2][2];
    auto mma = [&](auto HA) {
        constexpr int ha = decltype(HA)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hb = 0; hb < 2; ++hb)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2)
                        acc[ha][hb][f][g2] =
                            __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[hb][g2][ks], xa[f][ks], acc[ha][hb][f][g2], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto sync_fetch = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
        barrier_raw();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync_mma = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        barrier_raw();
    };

    // ---- bias -> LDS once per workgroup
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < p.N / 4; i += 512)
            *reinterpret_cast<f32x4*>(smem + BIAS_OFF + i * 16) = p.bias ? reinterpret_cast<const f32x4*>(p.bias)[i] : z;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // ---- T layout of this lane: r0 = lane & 3, i = (lane >> 2) & 3, q = lane >> 4; element offset (inside a tile) of slot 0 of
    // block (ha = 0, f = 0): row wm*32 + r0, column wn*64 + 16q + 4i
    const int r16 = lane & 15, q = lane >> 4, ti = (lane >> 2) & 3, r0 = lane & 3;
    const int ldc = p.ldc;
    const int t_off = (wm * 32 + r0) * ldc + wn * 64 + 16 * q + 4 * ti;
    float* const xg = reinterpret_cast<float*>(p.out);
    // parked / prefetched fp32 rows: pk[(ha*2 + f)*4 + j] = row ha*64 + wm*32 + f*16 + r0 + 4j of the tile, columns as above
    f32x4 pk[16];
    auto pk_off = [&](int idx) { return ((idx >> 3) * 64 + ((idx >> 2) & 1) * 16 + (idx & 3) * 4) * ldc; };
    // element offset of the previous tile's origin (its parked rows' home).  Every tile runs the same operations (the counted
    // waits need a fixed number of them): the FIRST tile "parks" its own untouched rows - K-tiles 1-4 write them back as they
    // are, K-tiles 5-8 read them again.
    size_t prev_base;
    {
        int tm, tn;
        tile_of(slot, tm, tn);
        prev_base = (size_t)tm * BM * ldc + tn * 256;
#pragma unroll
        for (int k = 0; k < 16; ++k) pk[k] = *reinterpret_cast<const f32x4*>(xg + prev_base + t_off + pk_off(k));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- prologue: W(0) A(0) W(1) A(1) W(2); A(2) is issued by the first PA, W(3) by the first PB
    issue_W(WDYN{}); issue_A(WDYN{});
    issue_W(WDYN{}); issue_A(WDYN{});
    issue_W(WDYN{});
    wait_vm<NWT>();
    barrier_raw();
    const bool late = wave >= 4;
    if (late) barrier_raw();

    int stg = 0;
    for (int r = 0; r < my_tiles; ++r) {
        int tm, tn;
        tile_of(slot + r * cpx, tm, tn);
        const int m0 = tm * BM, n0 = tn * 256;
        const size_t cur_base = (size_t)m0 * ldc + n0;
        const bool parked = r > 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) acc[a][b][f][g2] = f32x4{0.f, 0.f, 0.f, 0.f};
        float mun[2][2];                       // centre of row (ha, f, r16) for the fp16 copy (natural layout)
        // One K-tile.  KIND: 0 middle, 1 first of a tile, 2 / 3 / 4 the last three.  SP / SC: trickle operations issued in the
        // previous / this K-tile's PB.  OP: 0 none, 1 store pk[IDX .. IDX+3] (the PREVIOUS tile's updated rows), 2 load them
        // (THIS tile's residual rows).
        auto ktile = [&](auto KIND_T, auto SP_T, auto SC_T, auto OP_T, auto IDX_T) {
            constexpr int KIND = decltype(KIND_T)::value, SP = decltype(SP_T)::value, SC = decltype(SC_T)::value;
            constexpr int OP = decltype(OP_T)::value, IDX = decltype(IDX_T)::value;
            static_assert(OP != 0 || SC == 0, "trickle count");
            const int st = stg * STAGE;
            stg = stg == NST - 1 ? 0 : stg + 1;
            const bool more = KIND < 2 || r + 1 < my_tiles;
            (void)more;
            // ---------------- PA: fetch A0, W0, W1; refill A(g+2); quadrants (A0, .)
            read_A(0, st);
            read_W(st);
            if (KIND < 3 || more) issue_A(std::integral_constant<int, KIND == 3 ? 1 : 0>{});
            if constexpr (KIND == 4) {
#pragma unroll
                for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                    for (int f = 0; f < 2; ++f) mun[ha][f] = p.mu[m0 + ha * 64 + wm * 32 + f * 16 + r16];
            }
            sync_fetch();
            mma(I0{});
            sync_mma();
            // ---------------- PB: fetch A1; trickle; refill W(g+3); wait for A,W(g+1); quadrants (A1, .)
            read_A(1, st);
            if constexpr (OP == 1) {
#pragma unroll
                for (int k = 0; k < SC; ++k) *reinterpret_cast<f32x4*>(xg + prev_base + t_off + pk_off(IDX + k)) = pk[IDX + k];
            }
            if constexpr (OP == 2) {
#pragma unroll
                for (int k = 0; k < SC; ++k) pk[IDX + k] = *reinterpret_cast<const f32x4*>(xg + cur_base + t_off + pk_off(IDX + k));
            }
            if (KIND < 2 || more) issue_W(std::integral_constant<int, KIND == 2 ? 1 : 0>{});
            if constexpr (KIND == 0) wait_vm<NWT + SP + SC>();
            else if constexpr (KIND == 1) { if (parked) wait_vm<NWT + E>(); else wait_vm<NWT>(); }
            else if constexpr (KIND == 4) { if (more) wait_vm<NWT + R>(); }
            else if constexpr (KIND == 2) { if (more) wait_vm<NWT + SP>(); else wait_vm<0>(); }
            else { if (more) wait_vm<NWT>(); else wait_vm<0>(); }
            sync_fetch();
            mma(I1{});
            sync_mma();
        };
        {
            using Z = std::integral_constant<int, 0>;
            using C4 = std::integral_constant<int, 4>;
            using C5 = std::integral_constant<int, 5>;
            using C6 = std::integral_constant<int, 6>;
            using K0 = std::integral_constant<int, 0>;
            using K1 = std::integral_constant<int, 1>;
            using K2 = std::integral_constant<int, 2>;
            using K3 = std::integral_constant<int, 3>;
            using K4 = std::integral_constant<int, 4>;
            using ST = std::integral_constant<int, 1>;
            using LD = std::integral_constant<int, 2>;
            ktile(K1{}, Z{}, Z{}, Z{}, Z{});
            // K-tiles 1-4: the previous tile's rows leave (4 stores each)
            ktile(K0{}, Z{}, C4{}, ST{}, std::integral_constant<int, 0>{});
            ktile(K0{}, C4{}, C4{}, ST{}, std::integral_constant<int, 4>{});
            ktile(K0{}, C4{}, C4{}, ST{}, std::integral_constant<int, 8>{});
            ktile(K0{}, C4{}, C4{}, ST{}, std::integral_constant<int, 12>{});
            // K-tiles 5-7: this tile's residual rows arrive (6 + 5 + 5 loads); K-tile 8 only accounts for them
            ktile(K0{}, C4{}, C6{}, LD{}, std::integral_constant<int, 0>{});
            ktile(K0{}, C6{}, C5{}, LD{}, std::integral_constant<int, 6>{});
            ktile(K0{}, C5{}, C5{}, LD{}, std::integral_constant<int, 11>{});
            ktile(K0{}, C5{}, Z{}, Z{}, Z{});
            for (int kt = 9; kt < nk - 3; ++kt) ktile(K0{}, Z{}, Z{}, Z{}, Z{});
            ktile(K2{}, Z{}, Z{}, Z{}, Z{});
            ktile(K3{}, Z{}, Z{}, Z{}, Z{});
            ktile(K4{}, Z{}, Z{}, Z{}, Z{});
        }
        // ---------------- tile boundary: update in the T layout, park the fp32 rows, fp16 copy + statistics leave now
#ifdef HG_R2T_NOEPI      // timing experiment: no transposes / statistics (wrong results): what is the memory side worth?
#pragma unroll
        for (int k = 0; k < 16; ++k) pk[k] = pk[k] + acc[k >> 3][(k >> 2) & 1][(k >> 1) & 1][k & 1] + mun[0][0];
        if (false)
#endif
        {
            const int sg = tn * 4 + wn;
            const f32x4 bT = *reinterpret_cast<const f32x4*>(smem + BIAS_OFF + (n0 + wn * 64 + 16 * q + 4 * ti) * 4);
#pragma unroll
            for (int ha = 0; ha < 2; ++ha)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int blk = ha * 2 + f;
                    const int mb = m0 + ha * 64 + wm * 32 + f * 16;          // first row of the 16-row block
                    // accumulators -> T layout, one dword of the four blocks at a time
                    f32x4 t4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v[4] = {acc[ha][0][f][0][e], acc[ha][0][f][1][e], acc[ha][1][f][0][e], acc[ha][1][f][1][e]};
                        xpose4(v);
#pragma unroll
                        for (int j = 0; j < 4; ++j) t4[j][e] = v[j];
                    }
                    float muT[4] = {mun[ha][f], mun[ha][f], mun[ha][f], mun[ha][f]};
                    xpose4(muT);                                              // slot j: the centre of row r0 + 4j
                    u32x2 h16[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 v = pk[blk * 4 + j] + (t4[j] + bT);
                        pk[blk * 4 + j] = v;
                        // per row and per wave column group (64 columns): (sum, sum of squared deviations from the group mean)
                        float sum = (v[0] + v[1]) + (v[2] + v[3]);
                        sum = sum_rows(sum_i(sum));
                        const float gm = sum * (1.0f / 64.0f);
                        float m2 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = v[e] - gm;
                            m2 = fmaf(d, d, m2);
                        }
                        m2 = sum_rows(sum_i(m2));
                        if (lane < 4)
                            *reinterpret_cast<f32x2*>(p.stats + ((size_t)(mb + r0 + 4 * j) * p.stats_ld + sg) * 2) = f32x2{sum, m2};
                        half4 hh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hh[e] = (half_t)(v[e] - muT[j]);
                        h16[j] = __builtin_bit_cast(u32x2, hh);
                    }
                    // fp16 copy: lanes i and i ^ 1 pair their slots (2t, 2t+1): even i ends up with 8 consecutive columns of
                    // row r0 + 8t, odd i of row r0 + 8t + 4 - eight lanes per 128-byte row
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const u32x2 A = h16[2 * t], B = h16[2 * t + 1];
                        const u32x4 o = {dpp_u<0x114, 0xA>(A[0], B[0]), dpp_u<0x114, 0xA>(A[1], B[1]),
                                         dpp_u<0x104, 0x5>(B[0], A[0]), dpp_u<0x104, 0x5>(B[1], A[1])};
                        const int row = mb + r0 + 8 * t + 4 * (ti & 1);
                        *reinterpret_cast<u32x4*>(p.out2 + (size_t)row * ldc + n0 + wn * 64 + 16 * q + 4 * (ti & ~1)) = o;
                    }
                }
        }
        prev_base = cur_base;
    }
    // the last tile's rows
#pragma unroll
    for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(xg + prev_base + t_off + pk_off(k)) = pk[k];
    if (!late) barrier_raw();   // balances the extra barrier of the late waves
#endif
}

bool gemm_ring2t_ok(const GemmArgs& a) {
    return gemm_ring2_ok(a) && a.M % 128 == 0 && a.K >= 768 && a.out2 && a.stats && a.mu && a.stats_ld == 4 * (a.N / 256) &&
           ((a.M / 128) * (a.N / 256)) >= 1;
}

hipError_t launch_gemm_ring2t(const GemmArgs& a, hipStream_t s) {
    constexpr int RING = 3 * 49152;
    const int LDS = RING + a.N * 4;
    if (LDS > 160 * 1024 || !gemm_ring2t_ok(a)) return hipErrorInvalidValue;
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    static int n_cu_d[HG_MAX_DEVICES];
    const int dev_i = current_device_index();
    bool& attr_set = attr_set_d[dev_i];
    int& n_cu = n_cu_d[dev_i];
    if (!attr_set) {
        n_cu = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring2t),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n_cu = prop.multiProcessorCount;
        attr_set = true;
    }
    const int tiles_m = a.M / 128, tiles_n = a.N / 256;
    const int n_tiles = tiles_m * tiles_n;
    const int grid = n_tiles < n_cu ? n_tiles : n_cu;
    const size_t a_bytes = (size_t)tiles_m * 128 * a.lda * 2;
    static const int mode = []() {
        const char* d = getenv("HG_RING_DELAY");
        return (d ? atoi(d) : 2200) << 8;
    }();
    static const int gsz_env = []() { const char* e = getenv("HG_RING_GSZ"); return e ? atoi(e) : 0; }();
    int gsz = gsz_env > 0 ? gsz_env : (int)((1536 * 1024) / ((size_t)512 * a.K));
    if (gsz < 3) gsz = 3;
    if (gsz > tiles_n) gsz = tiles_n;
    if (gsz_env <= 0) {
        const int ngroups = (tiles_n + gsz - 1) / gsz;
        gsz = (tiles_n + ngroups - 1) / ngroups;
    }
    hipLaunchKernelGGL(gemm_ring2t, dim3(grid), dim3(512), LDS, s, a, tiles_n, n_tiles, (unsigned)a_bytes, mode, gsz);
    return hipGetLastError();
}

}  // namespace hg
