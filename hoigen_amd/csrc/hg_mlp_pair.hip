// The MLP of a transformer block as ONE persistent launch (gfx950): c_fc (LayerNorm-folded, + QuickGELU) -> c_proj (+ residual, emits
// the next LayerNorm's operand and statistics), clipnet/model.py:173-177,185-188 (reference: x = x + mlp(ln_2(x))).
//
// One workgroup per CU runs BOTH GEMMs' tiles: the 256 x 256 two-phase ring body (hg_gemm_ring_body.h, epilogue EPI_LN_BIAS_QGELU_F16) on
// its c_fc tiles and the 128 x 256 ring2 body (hg_gemm_ring2_body.h, EPI_RESID_LN_F32 on the hi / lo stream) on its c_proj tiles - the
// same K loops, epilogues and bits as the two stand-alone launches.  What the launch boundary used to order is ordered per 256-row panel
// of the intermediate activation `fc` instead:
//
//   producer  a c_fc tile is stored with plain stores; once all eight waves' stores have retired - they are in the XCD's L2 then - wave 0
//             adds 1 to ready[panel] (deferred into the second K-tile of the workgroup's next tile, where a counted vmcnt wait and the
//             phase's barriers prove it: no drain of the operand ring);
//   consumer  a c_proj tile (128 rows: half a panel) starts only when ready[panel] == N_fc / 256 column tiles; wave 0 issues
//             the counter load six K-tiles before the previous tile ends and looks at it three K-tiles later, a barrier publishes the
//             answer; its A operand is fetched with sc1 loads (past this CU's vector L1, from the XCD's L2).
//
// Producers and consumers of a row panel run ON THE SAME XCD, and that is what makes plain stores a valid hand-off: an XCD's L2 is
// coherent for all its CUs, the L2s of different XCDs are not with each other (MI355X_MICROARCH.md "Correctness boundaries"; measured
// here: the same kernel with the consumers on other XCDs reads stale lines, and storing write-through for them costs 50 us per launch,
// profiles/r06_mlp_pair.txt).  "Same XCD" is not an assumption about placement: a workgroup reads its XCD from the hardware register
// (HW_REG_XCC_ID) and takes the next free work slot OF THAT XCD from a census counter; XCD x's slots own the row panels p = x (mod 8),
// their c_fc tiles and their c_proj tiles.  So results do not depend on where or when workgroups land.  Progress needs every slot
// taken, i.e. gridDim / 8 workgroups on each XCD - what one workgroup per CU on the whole chip gives (the launcher refuses other
// devices); a workgroup beyond its XCD's slots exits, and a wait that outlasts its bound (~0.6 s: a slot nobody took) sets *err - a
// host-mapped word the API turns into HG_ERR_HIP - and goes on with whatever is there: wrong results behind an error, never a hang.
// Within a workgroup every c_fc tile precedes the c_proj tiles that could wait for it, so resident workgroups always progress.
// (The launch wants the GPU to itself: two such launches that each hold part of the CUs wait for each other until the bound expires.
// Across the streams of one process hg_api.hip orders them (PairGate); across processes INTEGRATION.md says so, option mlp_pair = 0.)
//
// Tile order.  XCD x walks its panels p = x + 8 k in CHUNKS of `ch` panels: inside a chunk the c_fc tiles run column group by column
// group (4 column tiles: their W slices stay in the XCD's L2 while the chunk's A panels stream, as in the stand-alone kernel's list);
// all XCDs finish their k-th panels at about the same time.  c_proj: the two 128-row halves of a panel, the column tiles of a half side
// by side (they share the A panel through L2).  A workgroup runs all its c_fc tiles, then its c_proj tiles - the launch boundary's order
// minus the boundary: a workgroup with one c_fc tile fewer starts its c_proj tiles a tile time earlier, its first panels are long
// complete.  (c_fc in two or three segments with the c_proj tiles of a segment's panels behind the next segment's c_fc - `fc` read back
// sooner after it was written - was built and measured no faster in any chunking: profiles/r06_mlp_pair.txt.)
#include <stdio.h>
#include <stdlib.h>

#include "hg_gemm_ring_body.h"
#include "hg_gemm_ring2_body.h"

namespace hg {

constexpr int MLP_SPIN_LIMIT = 1 << 20;    // polls of a ready counter (~0.5 us each: s_sleep 16) before giving up
constexpr int MLP_NX = 8;                  // XCDs (HW_REG_XCC_ID 0 .. 7)
constexpr int MLP_CENSUS = 16;             // words in front of the ready counters: [0, 8) work slots taken per XCD

struct MlpGeom {      // where this workgroup works: its XCD (hardware id), its work slot there, slots per XCD (all / those that run c_fc), panels per chunk
    int xcd, cu, cpx, fcs, ch;
    int wave;         // this wave's index in the workgroup (the bodies rebuild their thread id from it: see thread_id())
};
// threadIdx.x without keeping v0 alive: wave index (scalar) * 64 + lane (v_mbcnt), opaque so that nothing derived from it outlives a segment
__device__ __forceinline__ int mlp_thread_id(int wave) {
    int t = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    return t;
}

// ---- c_fc: the XCD's list = its chunks one after the other (inside a chunk: column group, panel, column in the group); the first `fcs`
// slots of the XCD deal it among themselves (items cu, cu + fcs, ...); a segment is the item range [e0, e0 + n) of this slot.  fcs < the
// number of slots leaves the other workgroups of the XCD idle (polling, asleep) until the first panels are complete: as in the
// stand-alone kernel, 2364 tiles take ten rounds on 240 or on 256 workgroups, and the idle CUs' power buys clock for the others.
struct MlpFcSched {
#ifdef HG_PAIR_NOPUB      // (timing experiment: c_fc without the hand-off's atomics and K-tile kinds; only meaningful with HG_PAIR_ONLY=1)
    static constexpr bool PUBLISH = false, CONSUME = false;
#else
    static constexpr bool PUBLISH = true, CONSUME = false;
#endif
    int x, cu, fcs, ch, tn, cg, wave;   // XCD, slot, slots that run c_fc, panels per chunk, column tiles, column tiles per group
    int npx, nfc, last_pc, total;       // panels of this XCD, full chunks, panels of the partial chunk, items of this slot
    int e0, n;
    unsigned* ready;
    __device__ __forceinline__ MlpFcSched(const MlpGeom& g, int M, int N, unsigned* ready_) {
        x = g.xcd; cu = g.cu; fcs = g.fcs; ch = g.ch; wave = g.wave; ready = ready_;
        tn = N / 256;
#ifdef HG_FC_CG      // (experiment: column tiles per group of the c_fc order)
        cg = tn % HG_FC_CG == 0 ? HG_FC_CG : tn;
#else
        cg = tn % 4 == 0 ? 4 : (tn % 3 == 0 ? 3 : tn);
#endif
        const int pf = (M + 255) / 256;
        npx = pf > x ? (pf - x + MLP_NX - 1) / MLP_NX : 0;
        nfc = npx / ch;
        last_pc = npx - nfc * ch;
        total = count(npx * tn);
        e0 = 0; n = 0;
    }
    // items of this slot among the first `lim` list positions
    __device__ __forceinline__ int count(int lim) const { return cu < fcs && cu < lim ? (lim - cu + fcs - 1) / fcs : 0; }
    // items of the chunks 0 .. s
    __device__ __forceinline__ int upto(int s) const {
        const int pc = (s + 1) * ch;
        return count((pc < npx ? pc : npx) * tn);
    }
    __device__ __forceinline__ int n_items() const { return n; }
    // start stagger of the launch's first segment, as in the stand-alone kernel: slots that own one tile fewer than the fullest ones spend a
    // pseudo-random fraction of a tile time before their first tile (de-phased epilogue bursts: -11 us per launch)
    __device__ __forceinline__ int slack() const { return e0 == 0 ? (npx * tn + fcs - 1) / fcs - total : 0; }
    __device__ __forceinline__ void tile(int r, int& tm, int& tn_) const {
        const int L = cu + fcs * (e0 + r);          // position in the XCD's list
        const int per_c = ch * tn;
        int c = L / per_c, l = L - c * per_c, pc = ch;
        if (c >= nfc) { c = nfc; l = L - nfc * per_c; pc = last_pc; }
        const int per_g = pc * cg;                  // inside the chunk: column group, panel, column in the group
        const int grp = l / per_g, rem = l - grp * per_g;
        const int j = rem / cg;
        tn_ = grp * cg + (rem - j * cg);
        tm = x + MLP_NX * (c * ch + j);
    }
    __device__ __forceinline__ int thread_id() const { return mlp_thread_id(wave); }
    __device__ __forceinline__ void publish(int tm, int lane) const {
        if (lane == 0) __hip_atomic_fetch_add(ready + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// ---- c_proj: items cu, cu + cpx, ... of the XCD's list (its panel k, 128-row half, column tile); a segment is the item range [e0, e0 + n)
struct MlpProjSched {
    static constexpr bool PUBLISH = false, CONSUME = true;
    int x, cu, cpx, ch, tn, tx, total, wave;
    int e0, n;
    unsigned target;
    const unsigned* ready;
    int* err;
    __device__ __forceinline__ MlpProjSched(const MlpGeom& g, int M, int N, int n_fc, const unsigned* ready_, int* err_) {
        // (the list is dealt from the LAST slot down: the c_fc list's remainder goes to the first slots, this one's to the last ones - a slot
        // owns 10 + 4, 9 + 5 or, two per XCD, 10 + 5 tiles instead of 10 + 5 for twelve: a c_proj tile takes 2.2 c_fc tile times)
        x = g.xcd; cu = g.cpx - 1 - g.cu; cpx = g.cpx; ch = g.ch; wave = g.wave; ready = ready_; err = err_;
        tn = N / 256;
        const int pf = (M + 255) / 256, pq = (M + 127) / 128;
        const int npx = pf > x ? (pf - x + MLP_NX - 1) / MLP_NX : 0;
        // the chip's last panel may have one 128-row half only; it is the last panel of its XCD's list, its missing half the list's tail
        tx = tn * (2 * npx - ((npx > 0 && x + MLP_NX * (npx - 1) == pf - 1 && (pq & 1)) ? 1 : 0));
        total = cu < tx ? (tx - cu + cpx - 1) / cpx : 0;
        target = (unsigned)(n_fc / 256);      // one publish per c_fc tile of the panel
        e0 = 0; n = 0;
    }
    // items whose panel lies in the c_fc chunks 0 .. t
    __device__ __forceinline__ int upto(int t) const {
        int lim = 2 * tn * ch * (t + 1);
        lim = lim < tx ? lim : tx;
        return cu < lim ? (lim - cu + cpx - 1) / cpx : 0;
    }
    __device__ __forceinline__ int n_items() const { return n; }
    __device__ __forceinline__ int slack() const { return 0; }      // (the slots reach their c_proj tiles de-phased by their c_fc tiles)
    __device__ __forceinline__ void tile(int r, int& tm, int& tn_) const {
        const int i = cu + cpx * (e0 + r);
        const int k = i / (2 * tn), rem = i - k * 2 * tn;
        const int half = rem / tn;
        tn_ = rem - half * tn;
        tm = 2 * (x + MLP_NX * k) + half;
    }
    __device__ __forceinline__ int thread_id() const { return mlp_thread_id(wave); }
    __device__ __forceinline__ const unsigned* counter(int r) const {
        int tm, t2;
        tile(r, tm, t2);
        return ready + (tm >> 1);
    }
    // one relaxed agent-scope load (global_load_dword sc1): the value is looked at K-tiles later
    __device__ __forceinline__ unsigned poll_issue(int r) const {
        return __hip_atomic_load(counter(r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void spin(const unsigned* ctr) const {
        for (int it = 0; it < MLP_SPIN_LIMIT; ++it) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return;
            __builtin_amdgcn_s_sleep(16);
            // (a wait of this launch or an earlier one has already given up: the call is lost, do not sit out the bound tile after tile)
            if ((it & 1023) == 1023 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
        }
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // never a hang: flag it and go on
    }
    __device__ __forceinline__ void poll_finish(int r, unsigned v) const {
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) < target) spin(counter(r));
    }
    __device__ __forceinline__ void poll_blocking(int r) const { spin(counter(r)); }
};

// Kernel arguments: only what the two bodies read (a GemmArgs each would be 2 x 47 dwords of scalar registers: see the kernel's note)
struct MlpFcArgs {
    const half_t* A; const half_t* W; const float* bias; half_t* out; const float* cs; const float* mr;
    int lda, ldc, M, N, K;
    unsigned a_bytes;
};
struct MlpProjArgs {
    const half_t* A; const half_t* W; const float* bias; float* out; const float* mu; half_t* out2; float* stats; half_t* lo;
    const float* muc;
    int lda, ldc, ld2, M, N, K, stats_ld;
    unsigned a_bytes;
    // GS instances (the text tower: the next LayerNorm's weight rides in the activation copy, GemmArgs::gamma)
    const float* gamma;
    half_t* out3;
    int ld3;
};
// the launch's tail: what finalize_stats would do in a launch of its own (read from the kernel-argument segment where it is needed)
struct MlpFinArgs {
    float* mr;            // [M][2]; null = the tail is off (the caller launches finalize_stats)
    float* mu;            // [M]
    float* muc;           // [M] or null
    int* range_flag;      // host-mapped, or null
};
struct MlpPairArgs {
    MlpFcArgs fc;
    MlpProjArgs proj;
    MlpFinArgs fin;
    unsigned* ready;      // [mlp_pair_ready_words(M)] zeroed before the launch: the census words, a counter per 256-row panel (c_fc tiles
                          // stored), a counter per 128-row half (c_proj tiles stored: the tail)
    int* err;             // host-mapped
    int ch;               // 256-row panels of an XCD per chunk
    int fc_slots;         // slots per XCD that run c_fc tiles (the others start with c_proj, i.e. wait)
    int census_off;       // byte offset in dynamic LDS of the word through which a workgroup's waves learn their slot
    unsigned long long* dbg;   // timing experiments (-DHG_PAIR_EXP builds, HG_PAIR_DBG): s_memtime stamps per workgroup
    int only;             // timing experiments (-DHG_PAIR_EXP builds, HG_PAIR_ONLY): 1 = the c_fc tiles alone, 2 = the c_proj tiles alone
                          // (waiting for nothing); wrong results
};

// Scalar registers are the scarce resource of this kernel.  With a whole GemmArgs per body as kernel arguments (2 x 47 dwords) the
// allocator spills scalars, and once it has to spill more than a dozen the scheduler minimises register pressure in EVERY region: the
// c_fc epilogue came out as ONE dependent chain through a single pair of temporaries (242 s_nop against 69 in the stand-alone kernel,
// +30 us per launch).  Reading the arguments back through an opaque pointer instead loses their address space (flat_ instead of
// global_ accesses, which also count in lgkmcnt: every fetch segment behind an epilogue then waits for the tile's stores, +19 us), and
// a loop around the two bodies (c_fc in several segments with c_proj between them - built, and no faster in any order:
// profiles/r06_mlp_pair.txt) makes both problems worse.  Hence: compact by-value arguments, straight-line code, c_fc then c_proj.
// c_proj's arguments are read from the kernel-argument segment BEHIND the c_fc body (as kernel arguments proper they are loaded at the
// kernel's entry and sit in 30 scalar registers all through c_fc: with the text tower's three more - gamma, out3, ld3 - that is what
// tips the allocator over, see above).  Read back as integers the pointers have lost their address space; a cast of the BITS to an
// address-space-1 pointer gives it back (a cast of the generic pointer there and back is folded away and leaves flat_ accesses).
template <class T>
__device__ __forceinline__ T load_kernarg(unsigned offset) {
    static_assert(sizeof(T) % 4 == 0, "dword-sized arguments");
    const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const __attribute__((address_space(4))) unsigned* w = reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(ka + offset);
    unsigned t[sizeof(T) / 4];
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) t[i] = w[i];
    T o;
    __builtin_memcpy(&o, t, sizeof(T));
    return o;
}
template <class T>
__device__ __forceinline__ T* global_bits(T* p) {
    return (T*)(__attribute__((address_space(1))) T*)(unsigned long long)p;
}

template <int HL, bool GS>
__global__ __launch_bounds__(512, 2) void mlp_pair_kernel(const MlpPairArgs P_in) {
#if defined(__HIP_DEVICE_COMPILE__)
    // this workgroup's XCD (the hardware's word, not blockIdx's) and its work slot there
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const MlpPairArgs& P = P_in;
    MlpGeom geo;
    geo.xcd = (int)(__builtin_amdgcn_s_getreg(20 /* HW_REG_XCC_ID */ | (0 << 6) | ((4 - 1) << 11)) & (MLP_NX - 1));
    geo.cpx = (gridDim.x + MLP_NX - 1) / MLP_NX;      // (slots per XCD; a grid launched short - the API's fault-injection option - leaves a slot untaken)
    geo.wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    geo.ch = P.ch;
    geo.fcs = P.fc_slots < geo.cpx ? P.fc_slots : geo.cpx;
    if (threadIdx.x == 0)
        *reinterpret_cast<int*>(smem + P.census_off) =
            (int)__hip_atomic_fetch_add(P.ready + geo.xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    geo.cu = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + P.census_off));
    if (geo.cu >= geo.cpx) return;      // more workgroups on this XCD than it has slots: the others own all of its work
#ifdef HG_PAIR_EXP
    unsigned long long t_stamp[4] = {__builtin_amdgcn_s_memtime(), 0, 0, 0};
#endif
    {
        MlpFcSched sf(geo, P.fc.M, P.fc.N, P.ready + MLP_CENSUS);
        sf.e0 = 0; sf.n = sf.total;
#ifdef HG_PAIR_EXP
        if (P.only == 2) sf.n = 0;
#endif
        if (sf.n > 0) {
            GemmArgs a{};
            a.A = P.fc.A; a.W = P.fc.W; a.bias = P.fc.bias; a.out = P.fc.out; a.cs = P.fc.cs; a.mr = P.fc.mr;
            a.lda = P.fc.lda; a.ldc = P.fc.ldc; a.M = P.fc.M; a.N = P.fc.N; a.K = P.fc.K;
            gemm_ring_body<4, EPI_LN_BIAS_QGELU_F16, true>(a, P.fc.a_bytes, 3000 << 8, sf);
        }
#ifdef HG_PAIR_EXP
        t_stamp[1] = __builtin_amdgcn_s_memtime();
        t_stamp[3] = (unsigned long long)sf.n;
#endif
    }
    {
        const MlpProjArgs Q = load_kernarg<MlpProjArgs>((unsigned)__builtin_offsetof(MlpPairArgs, proj));      // (here, not at the kernel's entry)
        MlpProjSched sp(geo, Q.M, Q.N, Q.K, P.ready + MLP_CENSUS, P.err);
        sp.e0 = 0; sp.n = sp.total;
#ifdef HG_PAIR_EXP
        if (P.only == 2) sp.target = 0u;
        if (P.only == 1) sp.n = 0;
#endif
        if (sp.n > 0) {
            GemmArgs a{};
            a.A = global_bits(Q.A); a.W = global_bits(Q.W); a.bias = global_bits(Q.bias); a.out = global_bits(Q.out); a.mu = global_bits(Q.mu);
            a.out2 = global_bits(Q.out2); a.stats = global_bits(Q.stats); a.lo = global_bits(Q.lo); a.muc = global_bits(Q.muc);
            a.lda = Q.lda; a.ldc = Q.ldc; a.ld2 = Q.ld2; a.M = Q.M; a.N = Q.N; a.K = Q.K;
            a.stats_ld = Q.stats_ld;
            if constexpr (GS) { a.gamma = global_bits(Q.gamma); a.out3 = global_bits(Q.out3); a.ld3 = Q.ld3; }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            barrier_raw();      // every wave has left the c_fc body (its LDS image is dead)
            gemm_ring2_body<EPI_RESID_LN_F32, HL, GS>(a, a.N / 256, Q.a_bytes, 0, sp);
        }
        // ---- tail: the LayerNorm statistics of the updated rows (what finalize_stats does in a launch of its own).  The column tiles of
        // a 128-row half ran on workgroups of THIS XCD; each of them, with its tile's partial sums in the XCD's L2 (every wave has drained
        // its stores), adds 1 to the half's counter, and the one whose add comes last combines the rows' partial sums - the same
        // function, the same bits.  Nothing here touches the K loops: a workgroup does it once, behind its last tile.
        const MlpFinArgs F = load_kernarg<MlpFinArgs>((unsigned)__builtin_offsetof(MlpPairArgs, fin));
        if (F.mr != nullptr) {
            wait_vm<0>();
            __syncthreads();
            const int tid = mlp_thread_id(geo.wave);      // (not threadIdx.x: see thread_id())
            const int M = Q.M, nt = Q.stats_ld, tn = Q.N / 256;
            unsigned* const done = P.ready + MLP_CENSUS + (M + 255) / 256;
            int* const slot_word = reinterpret_cast<int*>(smem + P.census_off);
            float* const stats = global_bits(Q.stats);
            float* const f_mr = global_bits(F.mr);
            float* const f_mu = global_bits(F.mu);
            float* const f_muc = global_bits(F.muc);
            // all of this workgroup's counter adds at once (lane e of wave 0: tile e), their answers through LDS: one atomic round trip
            int* const lastw = reinterpret_cast<int*>(smem);      // (the ring's first bytes: the bodies are done)
            if (tid < sp.n) {
                int tm, t2;
                sp.tile(tid, tm, t2);
                lastw[tid] = (int)__hip_atomic_fetch_add(done + tm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tn - 1 ? tm : -1;
            }
            __syncthreads();
            (void)slot_word;
            for (int e = 0; e < sp.n; ++e) {
                const int tm = lastw[e];      // the row half this workgroup finalises, or -1
                const int m = tm * 128 + (tid & 127);
                if (tm >= 0 && tid < 128 && m < M)
                    finalize_stats_row<true>(stats + (size_t)m * nt * 2, f_mr, f_mu, f_muc, m, nt, 64, 0, global_bits(F.range_flag));
            }
        }
#ifdef HG_PAIR_EXP
        t_stamp[2] = __builtin_amdgcn_s_memtime();
        if (P.dbg && threadIdx.x == 0) {
            unsigned long long* d = P.dbg + (size_t)blockIdx.x * 8;
            d[0] = t_stamp[0]; d[1] = t_stamp[1]; d[2] = t_stamp[2]; d[3] = t_stamp[3]; d[4] = (unsigned long long)sp.n;
            d[5] = (unsigned long long)(geo.xcd * 256 + geo.cu);
        }
#endif
    }
#endif
}

// c_fc: a LayerNorm-folded QuickGELU GEMM the 256 x 256 ring kernel takes; c_proj: the LayerNorm-emitting residual GEMM on fc
bool mlp_pair_ok(const GemmArgs& fc, const GemmArgs& proj, int n_cu) {
    if (n_cu != 32 * MLP_NX) return false;      // one workgroup per CU must put gridDim / 8 workgroups on each of the 8 XCDs (MI355X: 8 x 32 CUs)
    if (!gemm_ln_ok(EPI_LN_BIAS_QGELU_F16, fc) || !gemm_ln_ok(EPI_RESID_LN_F32, proj)) return false;
    if (fc.M != proj.M || proj.K != fc.N || proj.lda != fc.ldc || (const void*)proj.A != fc.out) return false;
    if (fc.K < 5 * 64 || proj.K < 8 * 64) return false;                    // K-tile kinds of the two hand-off protocols
    if ((size_t)((fc.M + 255) / 256) * 256 * fc.ldc * 2 >= (1ull << 32)) return false;      // fc through one buffer descriptor
    if (fc.M < 8 * 256) return false;
    // (the text tower: the next LayerNorm's weight in the activation copy - beside the stream's hi half where the stream leaves as hi / lo)
    if (proj.gamma && proj.hl == 2 && (!proj.out3 || proj.ld3 < proj.N || (proj.ld3 % 8))) return false;
    return proj.hl == 0 || proj.hl == 2 || proj.hl == 3;
}

template <int HL, bool GS = false>
static hipError_t launch_pair_t(const MlpPairArgs& a, int grid, int lds, hipStream_t s) {
    static bool attr_set_d[HG_MAX_DEVICES] = {};
    bool& attr_set = attr_set_d[current_device_index()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_pair_kernel<HL, GS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
#ifdef HG_PAIR_COOP      // (experiment: a cooperative launch - the runtime's co-residency guarantee - and what it costs per launch)
    MlpPairArgs a2 = a;
    void* args[] = {&a2};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&mlp_pair_kernel<HL, GS>), dim3(grid), dim3(512), args, (unsigned)lds, s);
#else
    hipLaunchKernelGGL((mlp_pair_kernel<HL, GS>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
#endif
}

size_t mlp_pair_ready_words(int M) { return (size_t)MLP_CENSUS + (size_t)((M + 255) / 256) + (size_t)((M + 127) / 128); }

hipError_t launch_mlp_pair(const GemmArgs& fc, const GemmArgs& proj_in, unsigned* ready, int* err, int ch, int fc_slots, int n_cu,
                           hipStream_t s, float* fin_mr, float* fin_mu, float* fin_muc, int* range_flag, int grid_short) {
    GemmArgs proj = proj_in;
    if (!proj.ld2) proj.ld2 = proj.ldc;
    if (!mlp_pair_ok(fc, proj, n_cu) || !ready || !err || proj.ld2 % 8) return hipErrorInvalidValue;
    if (proj.hl && (!proj.lo || !proj.mu || ((proj.hl == 2 || proj.hl == 3) && !proj.muc))) return hipErrorInvalidValue;
    MlpPairArgs a{};
    a.fc = MlpFcArgs{fc.A, fc.W, fc.bias, (half_t*)fc.out, fc.cs, fc.mr, fc.lda, fc.ldc, fc.M, fc.N, fc.K,
                     (unsigned)((size_t)((fc.M + 255) / 256) * 256 * fc.lda * 2)};
    a.proj = MlpProjArgs{proj.A, proj.W, proj.bias, (float*)proj.out, proj.mu, proj.out2, proj.stats, proj.lo, proj.muc,
                         proj.lda, proj.ldc, proj.ld2, proj.M, proj.N, proj.K, proj.stats_ld,
                         (unsigned)((size_t)((proj.M + 127) / 128) * 128 * proj.lda * 2), proj.gamma, proj.out3, proj.ld3};
    a.fin = MlpFinArgs{fin_mr, fin_mu, fin_muc, range_flag};
    if (fin_mr && (!fin_mu || fin_mu != proj.mu)) return hipErrorInvalidValue;      // (the centre the copy is written with IS the previous mean)
    a.ready = ready; a.err = err;
    a.ch = ch < 1 ? 1 : (ch > 64 ? 64 : ch);
    a.fc_slots = fc_slots < 1 ? 1 : fc_slots;
#ifdef HG_PAIR_EXP
    static const int only_env = []() { const char* e = getenv("HG_PAIR_ONLY"); return e ? atoi(e) : 0; }();
    a.only = only_env;
#endif
    const int lds_fc = 2 * (2 * 4 * 4096 + 2 * 16384) + fc.N * 4 * 2 + 256 * 8;
    const int lds_proj = 3 * 49152 + proj.N * 4 * (proj.gamma ? 2 : 1);
    a.census_off = lds_fc > lds_proj ? lds_fc : lds_proj;      // (behind both bodies' LDS images: never overwritten)
    const int lds = a.census_off + 16;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const int grid = n_cu - (grid_short > 0 && grid_short < MLP_NX ? grid_short : 0);
#ifdef HG_PAIR_EXP
    static int dbg_left = []() { const char* e = getenv("HG_PAIR_DBG"); return e ? atoi(e) : 0; }();
    if (dbg_left > 0 && proj.hl == 2) {
        --dbg_left;
        unsigned long long* d = nullptr;
        if (hipMalloc(&d, (size_t)grid * 64) != hipSuccess) return hipErrorOutOfMemory;
        hipMemsetAsync(d, 0, (size_t)grid * 64, s);
        a.dbg = d;
        hipError_t e = launch_pair_t<2>(a, grid, lds, s);
        hipStreamSynchronize(s);
        unsigned long long* h = (unsigned long long*)malloc((size_t)grid * 64);
        hipMemcpy(h, d, (size_t)grid * 64, hipMemcpyDeviceToHost);
        // s_memtime is not synchronised across CUs: only differences inside a workgroup mean anything.  Group by (c_fc tiles, c_proj tiles).
        for (int nf = 0; nf <= 12; ++nf)
            for (int np = 0; np <= 8; ++np) {
                double f = 0, q = 0, fmin = 1e30, fmax = 0, qmin = 1e30, qmax = 0; int n = 0;
                for (int b = 0; b < grid; ++b) {
                    if (!h[b * 8] || (int)h[b * 8 + 3] != nf || (int)h[b * 8 + 4] != np) continue;
                    const double d1 = (double)(h[b * 8 + 1] - h[b * 8]), d2 = (double)(h[b * 8 + 2] - h[b * 8 + 1]);
                    f += d1; q += d2; ++n;
                    fmin = d1 < fmin ? d1 : fmin; fmax = d1 > fmax ? d1 : fmax; qmin = d2 < qmin ? d2 : qmin; qmax = d2 > qmax ? d2 : qmax;
                }
                if (n) fprintf(stderr, "[pair-dbg] %3d workgroups with %2d c_fc + %d c_proj tiles: c_fc phase mean %.0f (min %.0f max %.0f) = %.0f per tile | c_proj phase mean %.0f (min %.0f max %.0f) = %.0f per tile [s_memtime ticks]\n",
                               n, nf, np, f / n, fmin, fmax, nf ? f / n / nf : 0.0, q / n, qmin, qmax, np ? q / n / np : 0.0);
            }
        free(h);
        hipFree(d);
        return e;
    }
#endif
    if (proj.gamma) {
        switch (proj.hl) {
            case 2: return launch_pair_t<2, true>(a, grid, lds, s);
            case 3: return launch_pair_t<3, true>(a, grid, lds, s);
            default: return hipErrorInvalidValue;
        }
    }
    switch (proj.hl) {
        case 0: return launch_pair_t<0>(a, grid, lds, s);
        case 2: return launch_pair_t<2>(a, grid, lds, s);
        case 3: return launch_pair_t<3>(a, grid, lds, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace hg
