// pdp_lqr_stream_kernels.h - LQR.lqrSolver (reference PDP/PDP.py:557-608) for dense matrices that live in HBM, as a RUNNER / STREAMER pair of
// wavefronts per trajectory: lqr_solve_stream_kernel<M>.  Same inputs, outputs and arithmetic as lqr_solve_kernel<M, 1> (pdp_lqr_kernels.h).
//
// Why.  lqr_solve_kernel streams its operand tiles straight into registers, one time step ahead of the Riccati step that uses them.  One
// step of compute (~2400 cycles backward, ~1000 forward) is shorter than an HBM round trip under load, the kernel needs 316 registers (one
// wave per SIMD), so every step waits for memory: 5.3 k cycles per step at C3 sizes, 1.85 TB/s.  Here the memory traffic is a second wave's
// job: the STREAMER copies the dense matrices of the next steps - 5.8 KB per step at C3 sizes, fully coalesced 512-byte lines, three steps
// in flight - into a ring of LDS slots; the RUNNER gathers its tiles from the ring (LDS latency, not HBM latency) and does nothing but the
// recursion.  They meet at two LDS counters (steps produced / consumed).  Workgroup = 512 threads = 4 trajectories (waves w and w + 4 share
// a SIMD, as in pdp_fused3_kernels.h), 40 KB of LDS per trajectory: Riccati scratch | counters | 4 slots of 1024 doubles.
// Applies when one parameter tile suffices (p <= 16 - m), n > 4 and a step's matrices fit a slot; everything else takes lqr_solve_kernel.
#pragma once
#include "pdp_lqr_kernels.h"
#include <type_traits>

namespace pdp {

constexpr int LQS_D = 4, LQS_LMAX = 16;                              // ring depth, most 64-double lines per slot (kernel template parameter NL <= LQS_LMAX: doubles per slot = 64 NL)
constexpr int LQS_SLICE = 160 * 1024 / 8 / 4;                        // doubles of LDS per trajectory
constexpr int LQS_MISC = RICCATI_SCRATCH, LQS_RING = LQS_MISC + 8;
static_assert(LQS_RING + LQS_D * 64 * LQS_LMAX <= LQS_SLICE, "ring does not fit the trajectory's LDS slice");

// segment offsets (doubles) inside a slot: backward sweep F | G | E | Hxx | Hxu | Hxe | Huu | Hue | 0.0, forward sweep F | G | E | K',k | P,W | 0.0
struct LqsLayout { int F, G, E, Hxx, Hxu, Hxe, Huu, Hue, Z, fF, fG, fE, fK, fPW, fZ; };
// P_{t+1}, W_{t+1} in the workspace of this kernel: the UPPER TRIANGLE of P (it is symmetrised every step: the two halves are bit-identical), row-major packed,
// followed by W [n][p] - 91 + 117 doubles per stage at C3 sizes instead of 169 + 117 (64 MB less per launch in each direction)
__host__ __device__ inline int lqs_ptri(int n, int i, int j) { return i <= j ? i * n - i * (i - 1) / 2 + (j - i) : j * n - j * (j - 1) / 2 + (i - j); }
__host__ __device__ inline int lqs_pw_doubles(int n, int p) { return n * (n + 1) / 2 + n * p; }
__host__ __device__ inline LqsLayout lqs_layout(int n, int m, int p, bool costate) {
    LqsLayout L;
    int o = 0;
    L.F = o; o += n * n; L.G = o; o += n * m; L.E = o; o += n * p; L.Hxx = o; o += n * n; L.Hxu = o; o += n * m; L.Hxe = o; o += n * p;
    L.Huu = o; o += m * m; L.Hue = o; o += m * p; L.Z = o;
    o = 0;
    L.fF = o; o += n * n; L.fG = o; o += n * m; L.fE = o; o += n * p; L.fK = o; o += n * m + m * p; L.fPW = o; o += costate ? lqs_pw_doubles(n, p) : 0; L.fZ = o;
    return L;
}
// lines per slot a problem needs (the kernel is instantiated for 12 and 16: the streamer issues NL loads and NL LDS stores per step whatever the data volume,
// and its three register sets are 3 NL doubles - at C3 sizes a backward step is 728 doubles, a forward step 634)
__host__ __device__ inline int lqs_lines(int n, int m, int p, bool costate) {
    const LqsLayout L = lqs_layout(n, m, p, costate);
    const int need = (L.Z > L.fZ ? L.Z : L.fZ) + 1;              // (+ the zero word behind the data)
    return (need + 63) / 64;
}
__host__ __device__ inline bool lqs_ok(int n, int m, int p, bool costate) {
    return n > 4 && n <= 16 && m <= 4 && p <= 16 - m && lqs_lines(n, m, p, costate) <= LQS_LMAX;
}

// Two kinds of signal.  lqs_signal_lds orders only the wave's LDS traffic in front of the counter (s_waitcnt lgkmcnt(0)): a RELEASE store would also
// wait for vmcnt(0) - for the streamer that is every load it has in flight for the NEXT steps (the pipeline would collapse to one step in flight),
// for the runner the HBM round trip of the gains it has just stored.  lqs_signal (release) is used once, where global memory really is handed over.
PDP_DEV void lqs_signal_lds(int* f, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
PDP_DEV void lqs_signal(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#ifdef PDP_LQS_TIMING      // probe builds (probes/lqr_stream_timing.py): cycles spent waiting at the counters
#define LQS_TW0() const long long tw0_ = __builtin_readcyclecounter()
#define LQS_TW1() lqs_waited += __builtin_readcyclecounter() - tw0_
static __device__ long long g_lqs_stamp[8];
#else
#define LQS_TW0()
#define LQS_TW1()
#endif
PDP_DEV void lqs_wait_ge(int* f, int v) {              // LDS hand-over only (LDS operations of a wave complete in order)
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// the runner's view of `produced`: the counter only grows, so the last value read is a lower bound - the LDS round trip of a poll is paid only when that
// bound does not already cover the step (the streamer is usually several steps ahead)
PDP_DEV void lqs_wait_cached(int* f, int v, int& seen) {
    while (seen < v) {
        seen = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (seen < v) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}
PDP_DEV void lqs_wait_ge_acquire(int* f, int v) {      // global memory handed over as well
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1);
}
PDP_DEV unsigned lqs_lds_addr(const double* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const double*)p; }

// offsets (doubles, inside a slot) of the four elements of a tile: from block A (at segA) where its map has the element, else from block B, else the zero
struct LqsGather { unsigned off[4]; };      // BYTE offsets
PDP_DEV LqsGather lqs_gather(int segA, const TileMap& mA, int segB, const TileMap& mB, int zoff) {
    LqsGather g;
#pragma unroll
    for (int r = 0; r < 4; ++r) g.off[r] = 8u * (unsigned)(mA.off[r] >= 0 ? segA + mA.off[r] : (mB.off[r] >= 0 ? segB + mB.off[r] : zoff));
    return g;
}
template <int NR = 4>
PDP_DEV d4 lqs_read(const LqsGather& g, unsigned slot_addr) {
    d4 v = zero4();
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = *(__attribute__((address_space(3))) const double*)(uintptr_t)(slot_addr + g.off[r]);
    return v;
}

// (range-checked buffer stores lqs_store / StoreMap / LQS_RSRC: pdp_lqr_kernels.h)

template <int M, int NL>
__global__ void __launch_bounds__(512) lqr_solve_stream_kernel(pdp_lqr_problem pr, double* __restrict__ Xo, double* __restrict__ Uo,
                                                               double* __restrict__ Lo, int32_t* __restrict__ status,
                                                               double* __restrict__ ws_gain, double* __restrict__ ws_pw) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = wid & 3;
    const bool runner = wid < 4;
    const int b = blockIdx.x * 4 + slot;
    double* lds = lds_all + slot * LQS_SLICE;
    double* scratch = lds;
    int* fl = (int*)(lds + LQS_MISC);                       // [0] steps produced [1] steps consumed (backward steps 0..T-1, forward steps T..2T-1)
    double* ring = lds + LQS_RING;
    if (runner && lane < 8) fl[lane] = 0;
    __syncthreads();
    if (b >= pr.B) return;
    const int n = pr.n, p = pr.p, T = pr.T;
    const bool costate = Lo != nullptr && ws_pw != nullptr;
    const LqsLayout L = lqs_layout(n, M, p, costate);
    constexpr int LQS_SP = 64 * NL;                          // doubles per ring slot
    const int gsz = n * M + M * p, pwsz = lqs_pw_doubles(n, p), ntri = n * (n + 1) / 2;
    const unsigned ring_addr = lqs_lds_addr(ring);
    const d4 z = zero4();
#ifdef PDP_LQS_TIMING
    long long lqs_waited = 0;
    const long long lqs_t0 = __builtin_readcyclecounter();
#endif

    if (!runner) {
        // ============================== streamer: HBM -> LDS ring, three steps in flight ==============================
        __builtin_amdgcn_s_setprio(0);
        struct Fam { pdp_mat m; int cnt; };
        const pdp_mat gains = {ws_gain, (int64_t)T * gsz, gsz}, pw = {costate ? ws_pw : nullptr, (int64_t)T * pwsz, pwsz};
        const Fam bw[8] = {{pr.F, n * n}, {pr.G, n * M}, {pr.E, n * p}, {pr.Hxx, n * n}, {pr.Hxu, n * M}, {pr.Hxe, n * p}, {pr.Huu, M * M}, {pr.Hue, M * p}};
        const Fam fw[5] = {{pr.F, n * n}, {pr.G, n * M}, {pr.E, n * p}, {gains, gsz}, {pw, costate ? pwsz : 0}};
        int seen_c = 0;                  // lower bound of fl[1] (steps consumed)
        const double* lp[NL];            // this lane's element of line l at the current step ...
        int lstep[NL];                   // ... and its byte distance to the same element of the next step (0: absent family / padding -> reads the zero word)
        auto phase = [&](const Fam* fam, int nfam, int zoff, int t_first, int dir, int kbase) {
            (void)zoff;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const int e = 64 * l + lane;
                lp[l] = PDP_ZERO; lstep[l] = 0;
                int o = 0;
                for (int f = 0; f < nfam; ++f) {
                    if (fam[f].m.ptr && e >= o && e < o + fam[f].cnt) { lp[l] = mat_at(fam[f].m, b, t_first) + (e - o); lstep[l] = dir * (int)(fam[f].m.tstride * 8); }
                    o += fam[f].cnt;
                }
            }
            double v0[NL], v1[NL];
            [[maybe_unused]] double v2[NL > 12 ? 1 : NL];       // slots of more than 12 lines: two register sets (three would not fit 256 registers beside the runner's code)
            auto issue = [&](double* v) {
#pragma unroll
                for (int l = 0; l < NL; ++l) { v[l] = *lp[l]; lp[l] = (const double*)((const char*)lp[l] + lstep[l]); }      // no branches in here:
                // control flow between the loads makes the compiler's s_waitcnt placement conservative (vmcnt(0) at every join), which would
                // serialise the steps in flight; lines past the step's data read the zero word and land in the slot's padding
            };
            auto commit = [&](const double* v, int j) {
                const int k = kbase + j;
                { LQS_TW0(); lqs_wait_cached(fl + 1, k - LQS_D + 1, seen_c); LQS_TW1(); }      // the slot's previous occupant has been consumed
                double* s = ring + (j % LQS_D) * LQS_SP;              // slot = step index inside the phase, mod D (both phases start on slot 0)
#pragma unroll
                for (int l = 0; l < NL; ++l) s[64 * l + lane] = v[l];
                lqs_signal_lds(fl + 0, k + 1);
            };
            issue(v0);
            if constexpr (NL > 12) {                         // two steps in flight (2 x 8 KB per trajectory)
                for (int j = 0; j < T; j += 2) {
                    if (j + 1 < T) issue(v1);
                    commit(v0, j);
                    if (j + 1 < T) { if (j + 2 < T) issue(v0); commit(v1, j + 1); }
                }
            } else {                                         // three steps in flight (3 x 6 KB)
                if (T > 1) issue(v1);
                for (int j = 0; j < T; j += 3) {
                    if (j + 2 < T) issue(v2);
                    commit(v0, j);
                    if (j + 1 < T) { if (j + 3 < T) issue(v0); commit(v1, j + 1); }
                    if (j + 2 < T) { if (j + 4 < T) issue(v1); commit(v2, j + 2); }
                }
            }
        };
        phase(bw, 8, L.Z, T - 1, -1, 0);
        lqs_wait_ge_acquire(fl + 1, T);                      // gains (and P, W) of every step are in memory: the runner released them with its last signal
        phase(fw, 5, L.fZ, 0, 1, T);
#ifdef PDP_LQS_TIMING
        if (b == 0 && lane == 0) { g_lqs_stamp[4] = __builtin_readcyclecounter() - lqs_t0; g_lqs_stamp[5] = lqs_waited; }
#endif
        return;
    }

    // ======================================= runner: the recursion, operands from the ring =======================================
    __builtin_amdgcn_s_setprio(3);
    const int p0 = p;
    bool ok = true, finite = true;
    int seen = 0;                                           // lower bound of fl[0] (steps produced)
    const TileMap mNN = make_dense_map<false>(n, n, n, 0, 0, lane), mNM = make_dense_map<false>(n, M, M, 0, 0, lane),
                  mNP = make_dense_map<false>(n, p0, p, 0, M, lane), mMM = make_dense_map<false>(M, M, M, 0, 0, lane),
                  mMP = make_dense_map<false>(M, p0, p, 0, M, lane), mFT = make_dense_map<true>(n, n, n, 0, 0, lane),
                  mGT = make_dense_map<true>(n, M, M, 0, 0, lane), mNMrep = make_rep4_map(n, M, M, lane);
    const TileMap mNone = {{-1, -1, -1, -1}};
    TileMap mPst, mPld;                                      // P in the workspace: stored from the upper triangle of the tile, read back symmetrically
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tile_row(lane, r), col = tile_col(lane);
        mPst.off[r] = (row < n && col < n && row <= col) ? lqs_ptri(n, row, col) : -1;
        mPld.off[r] = (row < n && col < n) ? lqs_ptri(n, row, col) : -1;
    }
    // terminal condition: PP[T-1] = hxx, WW[T-1] = hxe (PDP.py:561-562)
    d4 P = load_map(mat_at(pr.hxx, b, 0), mNN);
    d4 W0;
    {
        const double* hxe = mat_at(pr.hxe, b, 0);
        W0 = hxe ? load_map(hxe, mNP) : z;
    }
    {
        const LqsGather gF = lqs_gather(L.F, mNN, 0, mNone, L.Z), gY = lqs_gather(L.G, mNM, L.E, mNP, L.Z), gGr = lqs_gather(L.G, mNMrep, 0, mNone, L.Z),
                        gHxx = lqs_gather(L.Hxx, mNN, 0, mNone, L.Z), gHX = lqs_gather(L.Hxu, mNM, L.Hxe, mNP, L.Z),
                        gHU = lqs_gather(L.Huu, mMM, L.Hue, mMP, L.Z), gHux = lqs_gather(L.Hxu, mGT, 0, mNone, L.Z);
        constexpr unsigned SB = 8u * LQS_SP;                 // bytes per slot: the slot of a step is a literal offset in the 4-step trips below
        const StoreMap sNN = lqs_store_map(mPst), sNP = lqs_store_map(mNP), sNM = lqs_store_map(mNM), sMP = lqs_store_map(mMP);
        const auto rPW = LQS_RSRC(ws_pw ? ws_pw + (int64_t)b * T * pwsz : ws_gain, ws_pw ? (int64_t)T * pwsz * 8 : 0);
        const auto rG = LQS_RSRC(ws_gain + (int64_t)b * T * gsz, (int64_t)T * gsz * 8);
        lqs_wait_cached(fl + 0, T > 1 ? 2 : 1, seen);
        d4 Fa = lqs_read(gF, ring_addr), Ya = lqs_read(gY, ring_addr), Fb = z, Yb = z;
        auto bstep = [&](int k, unsigned so, unsigned sn, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {      // so / sn: byte offsets of this / the next step's slot
            { LQS_TW0(); lqs_wait_cached(fl + 0, k + 2 < T ? k + 2 : T, seen); LQS_TW1(); }      // this step's slot and the next one's are filled
            const unsigned sa = ring_addr + so;
            d4 Hxx = lqs_read(gHxx, sa), HX2 = lqs_read(gHX, sa), HU2 = lqs_read<1>(gHU, sa), Grep = lqs_read(gGr, sa), Hux = lqs_read<1>(gHux, sa);
            if (k + 1 < T) { Fn = lqs_read(gF, ring_addr + sn); Yn = lqs_read(gY, ring_addr + sn); }
            const int t = T - 1 - k;
            lqs_store(rPW, sNN, (unsigned)(t * pwsz) * 8u, P);               // P_{t+1}, W_{t+1} for the costate output (lambda_{t+1} = P x_{t+1} + W, PDP.py:604)
            lqs_store(rPW, sNP, (unsigned)(t * pwsz + ntri) * 8u, W0);
            RiccatiGains g;
            d4 P_old;
            ok = riccati_backward<M, true, false>(P, W0, Fc, Yc, Grep, Hxx, HX2, HU2, Hux[0], scratch, lane, p0, g, P_old) && ok;
            lqs_store(rG, sNM, (unsigned)(t * gsz) * 8u, g.KT);
            lqs_store<1>(rG, sMP, (unsigned)(t * gsz + n * M) * 8u, g.IK);
            if (k + 1 < T) lqs_signal_lds(fl + 1, k + 1);   // the slot is free (its tiles are in registers)
            else lqs_signal(fl + 1, T);                       // last step: RELEASE - the gains / P, W of all steps are handed to the streamer
        };
        static_assert(LQS_D == 4, "the trips below are written for a ring of four slots");
        int k = 0;
        for (; k + 3 < T; k += 4) {
            bstep(k, 0 * SB, 1 * SB, Fa, Ya, Fb, Yb); bstep(k + 1, 1 * SB, 2 * SB, Fb, Yb, Fa, Ya);
            bstep(k + 2, 2 * SB, 3 * SB, Fa, Ya, Fb, Yb); bstep(k + 3, 3 * SB, 0 * SB, Fb, Yb, Fa, Ya);
        }
        if (k < T) bstep(k, 0 * SB, 1 * SB, Fa, Ya, Fb, Yb);
        if (k + 1 < T) bstep(k + 1, 1 * SB, 2 * SB, Fb, Yb, Fa, Ya);
        if (k + 2 < T) bstep(k + 2, 2 * SB, 3 * SB, Fa, Ya, Fb, Yb);
        finite = finite && tile_finite(P) && tile_finite(W0);      // (a non-finite P or W propagates to the last step: one check at the end)
    }
    // ---- forward rollout (PDP.py:582-608)
    d4 X;
    {
        const double* X0 = mat_at(pr.X0, b, 0);
        X = X0 ? load_map(X0, mNP) : z;
        store_map(Xo + (int64_t)b * (T + 1) * n * p, mNP, X);
    }
    {
        const LqsGather gFT = lqs_gather(L.fF, mFT, 0, mNone, L.fZ), gGT = lqs_gather(L.fG, mGT, 0, mNone, L.fZ), gKT = lqs_gather(L.fK, mNMrep, 0, mNone, L.fZ),
                        gk = lqs_gather(L.fK + n * M, mMP, 0, mNone, L.fZ), gE = lqs_gather(L.fE, mNP, 0, mNone, L.fZ),
                        gP = lqs_gather(L.fPW, costate ? mPld : mNone, 0, mNone, L.fZ), gW = lqs_gather(L.fPW + ntri, costate ? mNP : mNone, 0, mNone, L.fZ);
        constexpr unsigned SB = 8u * LQS_SP;
        const StoreMap sNP = lqs_store_map(mNP), sMP = lqs_store_map(mMP);
        const auto rU = LQS_RSRC(Uo + (int64_t)b * T * M * p, (int64_t)T * M * p * 8), rX = LQS_RSRC(Xo + (int64_t)b * (T + 1) * n * p, (int64_t)(T + 1) * n * p * 8),
                   rL = LQS_RSRC(Lo ? Lo + (int64_t)b * T * n * p : Uo, Lo ? (int64_t)T * n * p * 8 : 0);
        struct FwdTiles { d4 FT, GT, KT, Pt, k, Et, Wt; };
        auto load_fwd = [&](FwdTiles& w, unsigned so) {
            const unsigned sa = ring_addr + so;
            w.FT = lqs_read(gFT, sa); w.GT = lqs_read<1>(gGT, sa); w.KT = lqs_read(gKT, sa); w.k = lqs_read<1>(gk, sa); w.Et = lqs_read(gE, sa);
            w.Pt = lqs_read(gP, sa); w.Wt = lqs_read(gW, sa);
        };
        auto fstep = [&](int j, unsigned sn, const FwdTiles& c, FwdTiles& nx) {      // sn: byte offset of the NEXT step's slot
            if (j + 1 < T) { { LQS_TW0(); lqs_wait_cached(fl + 0, T + j + 2, seen); LQS_TW1(); } load_fwd(nx, sn); }
            const d4 KTn = -c.KT, kn = -c.k;
            d4 U, Xn;
            riccati_forward(KTn, kn, c.FT, c.GT, c.Et, X, U, Xn);
            X = Xn;
            lqs_store<1>(rU, sMP, (unsigned)(j * M * p) * 8u, U);
            lqs_store(rX, sNP, (unsigned)((j + 1) * n * p) * 8u, Xn);
            if (Lo) {
                d4 Lm = mma_tn(c.Pt, Xn, c.Wt);              // P x+ + W  (P symmetric)
                lqs_store(rL, sNP, (unsigned)(j * n * p) * 8u, Lm);
            }
            lqs_signal_lds(fl + 1, T + j + 1);
        };
        FwdTiles ta, tb;
#ifdef PDP_LQS_TIMING
        if (b == 0 && lane == 0) { g_lqs_stamp[1] = __builtin_readcyclecounter() - lqs_t0; g_lqs_stamp[2] = lqs_waited; }
#endif
        { LQS_TW0(); lqs_wait_cached(fl + 0, T + 1, seen); LQS_TW1(); }
        load_fwd(ta, 0u);
        int j = 0;
        for (; j + 3 < T; j += 4) { fstep(j, 1 * SB, ta, tb); fstep(j + 1, 2 * SB, tb, ta); fstep(j + 2, 3 * SB, ta, tb); fstep(j + 3, 0 * SB, tb, ta); }
        if (j < T) fstep(j, 1 * SB, ta, tb);
        if (j + 1 < T) fstep(j + 1, 2 * SB, tb, ta);
        if (j + 2 < T) fstep(j + 2, 3 * SB, ta, tb);
        finite = finite && tile_finite(X);
    }
    int st = 0;
    if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
    if (!ok) st |= PDP_STATUS_PIVOT;
    if (lane == 0 && status) status[b] = st;
#ifdef PDP_LQS_TIMING
    if (b == 0 && lane == 0) { g_lqs_stamp[0] = __builtin_readcyclecounter() - lqs_t0; g_lqs_stamp[3] = lqs_waited; }
#endif
}

#ifdef PDP_LQS_TIMING
__global__ void lqs_read_stamps(long long* out) { if (threadIdx.x < 8) out[threadIdx.x] = g_lqs_stamp[threadIdx.x]; }
#endif

}  // namespace pdp
