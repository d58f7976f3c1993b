// pdp_lqr_kernels.h - the batched LQR.lqrSolver kernel (reference PDP/PDP.py:557-608) and its operand streaming helpers.
// Shared by libpdp_hip.so (pdp_lqr_solve_batched) and the per-model libraries (the Newton solve of pdp_oc_solve_batched solves
// its LQ sub-problems with the same kernel).
#pragma once
#include "../../include/pdp_hip.h"
#include "pdp_riccati.h"
#include "pdp_riccati_small.h"

namespace pdp {

PDP_DEV const double* mat_at(const pdp_mat& M, int b, int t) {
    return M.ptr ? M.ptr + (int64_t)b * M.bstride + (int64_t)t * M.tstride : nullptr;
}

// Streaming operand tiles.  Each lane keeps, per tile register, a pointer to ITS element of the current time step and the byte
// stride to the same element of the next step; elements a tile does not have (padding, absent optional matrices) point at a zero
// word with stride 0.  A load is then the bare global_load - no bound check, select or add behind it - so the tiles of step t-1
// can be requested while step t computes and the wait lands at their first use (with post-processing next to the load the
// compiler waits for the data immediately and every step pays an HBM round trip: 2.4 us per step instead of ~1).
static __device__ double PDP_ZERO[2] = {0.0, 0.0};      // never written; not `const`: a constant-address-space zero would turn every streamed load
                                                           // into a FLAT load, which also counts on lgkmcnt and is waited for at each LDS hand-off
struct RunPtr { const double* p[4]; int step[4]; };
template <int NR = 4>
PDP_DEV RunPtr make_run(const pdp_mat& A, const TileMap& mA, const pdp_mat& Bm, const TileMap& mB, int b, int t) {
    RunPtr r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.p[k] = PDP_ZERO; r.step[k] = 0;
        if (k < NR) {
            if (A.ptr && mA.off[k] >= 0) { r.p[k] = mat_at(A, b, t) + mA.off[k]; r.step[k] = (int)(A.tstride * 8); }
            else if (Bm.ptr && mB.off[k] >= 0) { r.p[k] = mat_at(Bm, b, t) + mB.off[k]; r.step[k] = (int)(Bm.tstride * 8); }
        }
    }
    return r;
}
template <int NR = 4>
PDP_DEV d4 load_run(RunPtr& r, int dir) {          // read the current step's elements, then move by dir (+1 / -1) time steps
    d4 v = zero4();
#pragma unroll
    for (int k = 0; k < NR; ++k) { v[k] = *r.p[k]; r.p[k] = (const double*)((const char*)r.p[k] + dir * r.step[k]); }
    return v;
}

// Branch-free stores of tiles into exact-size arrays through BUFFER instructions: every lane keeps, per tile register, the byte offset of its element
// inside one time step's block - or 0x80000000 if the array has no such element, which the buffer's range check drops in hardware (num_records =
// the trajectory's bytes of that array, far below 2 GB; 0 for an absent array: everything dropped); the time step is the instruction's scalar offset.
// store_map's predicated stores cost a basic block each (mask reload, branch, 64-bit address arithmetic) - ~20 per backward step.
typedef unsigned lqs_u2 __attribute__((ext_vector_type(2)));
struct StoreMap { unsigned off[4]; };
PDP_DEV StoreMap lqs_store_map(const TileMap& m) {
    StoreMap r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.off[k] = m.off[k] >= 0 ? 8u * (unsigned)m.off[k] : 0x80000000u;
    return r;
}
#define LQS_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
template <int NR = 4, class R>
PDP_DEV void lqs_store(R rs, const StoreMap& m, unsigned soff, const d4 v) {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const double x = v[k];          // (bit-casting the vector element directly made every store write register 0's value)
        lqs_u2 w;
        w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x);
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, m.off[k], soff, 0);
    }
}

// gains workspace per (b,t): KT [n*m] then k [m*p];  P/W workspace per (b,t): P [n*n] then W [n*p]
template <int M, int NT>
__global__ void __launch_bounds__(64) lqr_solve_kernel(pdp_lqr_problem pr, double* __restrict__ Xo, double* __restrict__ Uo,
                                                        double* __restrict__ Lo, int32_t* __restrict__ status,
                                                        double* __restrict__ ws_gain, double* __restrict__ ws_pw) {
    __shared__ double scratch[RICCATI_SCRATCH];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = pr.n, p = pr.p, T = pr.T;
    const int p0 = min(p, 16 - M);
    const d4 z = zero4();
    const int gsz = n * M + M * p, pwsz = n * n + n * p;
    bool ok = true, finite = true;
    // loop-invariant tile <-> dense maps of the first parameter tile (parameter columns sit behind the M control columns)
    const TileMap mNN = make_dense_map<false>(n, n, n, 0, 0, lane), mNM = make_dense_map<false>(n, M, M, 0, 0, lane),
                  mNP = make_dense_map<false>(n, p0, p, 0, M, lane), mMM = make_dense_map<false>(M, M, M, 0, 0, lane),
                  mMP = make_dense_map<false>(M, p0, p, 0, M, lane), mFT = make_dense_map<true>(n, n, n, 0, 0, lane),
                  mGT = make_dense_map<true>(n, M, M, 0, 0, lane), mNMrep = make_rep4_map(n, M, M, lane);   // G / K' replicated in the 4 column blocks

    // first-parameter-tile outputs leave through range-checked buffer stores (one resource per array, this trajectory's part of it)
    const StoreMap sNN = lqs_store_map(mNN), sNP = lqs_store_map(mNP), sNM = lqs_store_map(mNM), sMP = lqs_store_map(mMP);
    const auto rPW = LQS_RSRC(ws_pw ? ws_pw + (int64_t)b * T * pwsz : ws_gain, ws_pw ? (int64_t)T * pwsz * 8 : 0);
    const auto rG = LQS_RSRC(ws_gain + (int64_t)b * T * gsz, (int64_t)T * gsz * 8);
    const auto rU = LQS_RSRC(Uo + (int64_t)b * T * M * p, (int64_t)T * M * p * 8), rX = LQS_RSRC(Xo + (int64_t)b * (T + 1) * n * p, (int64_t)(T + 1) * n * p * 8),
               rL = LQS_RSRC(Lo ? Lo + (int64_t)b * T * n * p : Uo, Lo ? (int64_t)T * n * p * 8 : 0);
    // terminal condition: PP[T-1] = hxx, WW[T-1] = hxe (PDP.py:561-562)
    d4 P = load_map(mat_at(pr.hxx, b, 0), mNN);
    d4 W[NT];
    {
        const double* hxe = mat_at(pr.hxe, b, 0);
        W[0] = hxe ? load_map(hxe, mNP) : z;
#pragma unroll
        for (int j = 1; j < NT; ++j) W[j] = hxe ? load_dense<false>(hxe + p0 + 16 * (j - 1), n, min(16, p - p0 - 16 * (j - 1)), p, 0, 0, lane) : z;
    }
    // The matrices of step t-1 are requested from HBM before the Riccati step of t runs (one step = ~3k cycles of MFMA / VALU work,
    // an HBM round trip ~2k): two steps per trip, the operand tiles alternating between two register sets.
    struct BwdTiles { d4 Ft, Y2, Grep, Hxx, HX2, HU2; };
    const pdp_mat none = {nullptr, 0, 0};
    RunPtr rF = make_run(pr.F, mNN, none, mNN, b, T - 1), rY = make_run(pr.G, mNM, pr.E, mNP, b, T - 1), rHxx = make_run(pr.Hxx, mNN, none, mNN, b, T - 1),
           rHX = make_run(pr.Hxu, mNM, pr.Hxe, mNP, b, T - 1), rHU = make_run<1>(pr.Huu, mMM, pr.Hue, mMP, b, T - 1),
           rGr = make_run(pr.G, mNMrep, none, mNN, b, T - 1);
    auto load_bwd = [&](BwdTiles& w) {     // (issued for steps t-1 >= 0 only: bstep guards the request of the last step)
        w.Ft = load_run(rF, -1); w.Y2 = load_run(rY, -1); w.Grep = load_run(rGr, -1); w.Hxx = load_run(rHxx, -1); w.HX2 = load_run(rHX, -1);
        w.HU2 = load_run<1>(rHU, -1);
    };
    auto bstep = [&](int t, const BwdTiles& c, BwdTiles& nx) {
        if (t > 0) load_bwd(nx);      // (never for t - 1 = -1: one time stride in front of the arrays - the round-1 out-of-bounds read, profiles/r02_lqr_oob_root_cause.txt)
        if (ws_pw) {   // P_{t+1}, W_{t+1} for the costate output (lambda_{t+1} = P x_{t+1} + W, PDP.py:604)
            double* pw = ws_pw + ((int64_t)b * T + t) * pwsz;
            lqs_store(rPW, sNN, (unsigned)(t * pwsz) * 8u, P);
            lqs_store(rPW, sNP, (unsigned)(t * pwsz + n * n) * 8u, W[0]);
#pragma unroll
            for (int j = 1; j < NT; ++j) store_dense(pw + n * n + p0 + 16 * (j - 1), n, min(16, p - p0 - 16 * (j - 1)), p, 0, 0, lane, W[j]);
        }
        RiccatiGains g;
        d4 P_old;
        ok = riccati_backward<M, true, true>(P, W[0], c.Ft, c.Y2, c.Grep, c.Hxx, c.HX2, c.HU2, 0.0, scratch, lane, p0, g, P_old) && ok;
        double* gw = ws_gain + ((int64_t)b * T + t) * gsz;
        lqs_store(rG, sNM, (unsigned)(t * gsz) * 8u, g.KT);
        lqs_store<1>(rG, sMP, (unsigned)(t * gsz + n * M) * 8u, g.IK);
        if constexpr (NT > 1) {
            const double *E = mat_at(pr.E, b, t), *Hxe = mat_at(pr.Hxe, b, t), *Hue = mat_at(pr.Hue, b, t);
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                const int c0 = p0 + 16 * (j - 1), w = min(16, p - c0);
                d4 Ej = E ? load_dense<false>(E + c0, n, w, p, 0, 0, lane) : z;
                d4 Hxej = Hxe ? load_dense<false>(Hxe + c0, n, w, p, 0, 0, lane) : z;
                d4 Huej = Hue ? load_dense<false>(Hue + c0, M, w, p, 0, 0, lane) : z;
                d4 kj;
                riccati_backward_extra(P_old, W[j], c.Ft, c.Grep, Ej, Hxej, Huej, g, kj);
                store_dense(gw + n * M + c0, M, w, p, 0, 0, lane, kj);
            }
        }
        finite = finite && tile_finite(P) && tile_finite(W[0]);
    };
    {
        BwdTiles ta, tb;
        load_bwd(ta);
        int t = T - 1;
        for (; t >= 1; t -= 2) { bstep(t, ta, tb); bstep(t - 1, tb, ta); }
        if (t == 0) bstep(0, ta, tb);
    }
    // ---- forward rollout (PDP.py:582-608)
    d4 X[NT];
    {
        const double* X0 = mat_at(pr.X0, b, 0);
        X[0] = X0 ? load_map(X0, mNP) : z;
#pragma unroll
        for (int j = 1; j < NT; ++j) X[j] = X0 ? load_dense<false>(X0 + p0 + 16 * (j - 1), n, min(16, p - p0 - 16 * (j - 1)), p, 0, 0, lane) : z;
        double* x0o = Xo + (int64_t)b * (T + 1) * n * p;
        store_map(x0o, mNP, X[0]);
#pragma unroll
        for (int j = 1; j < NT; ++j) store_dense(x0o + p0 + 16 * (j - 1), n, min(16, p - p0 - 16 * (j - 1)), p, 0, 0, lane, X[j]);
    }
    __threadfence_block();
    // forward sweep, same scheme: everything step t+1 reads for its first parameter tile is requested while step t computes
    struct FwdTiles { d4 FT, GT, KT, Pt, k, Et, Wt; };
    const pdp_mat gKT = {ws_gain, (int64_t)T * gsz, gsz}, gk = {ws_gain + n * M, (int64_t)T * gsz, gsz},
                  wP = {ws_pw, (int64_t)T * pwsz, pwsz}, wW = {(ws_pw && Lo) ? ws_pw + n * n : nullptr, (int64_t)T * pwsz, pwsz};
    RunPtr qFT = make_run(pr.F, mFT, none, mFT, b, 0), qGT = make_run<1>(pr.G, mGT, none, mGT, b, 0), qKT = make_run(gKT, mNMrep, none, mNM, b, 0),
           qk = make_run<1>(gk, mMP, none, mMP, b, 0), qE = make_run(pr.E, mNP, none, mNP, b, 0), qP = make_run(wP, mNN, none, mNN, b, 0),
           qW = make_run(wW, mNP, none, mNP, b, 0);
    auto load_fwd = [&](FwdTiles& w) {
        w.FT = load_run(qFT, 1); w.GT = load_run<1>(qGT, 1); w.KT = load_run(qKT, 1); w.k = load_run<1>(qk, 1);
        w.Et = load_run(qE, 1); w.Pt = load_run(qP, 1); w.Wt = load_run(qW, 1);
    };
    auto fstep = [&](int t, const FwdTiles& c, FwdTiles& nx) {
        if (t + 1 < T) load_fwd(nx);
        const d4 KTn = -c.KT;
        const double* E = mat_at(pr.E, b, t);
        const double* gw = ws_gain + ((int64_t)b * T + t) * gsz;
        const double* pw = ws_pw ? ws_pw + ((int64_t)b * T + t) * pwsz : nullptr;
        double* xo = Xo + ((int64_t)b * (T + 1) + t + 1) * n * p;
        double* uo = Uo + ((int64_t)b * T + t) * M * p;
        double* lo = Lo ? Lo + ((int64_t)b * T + t) * n * p : nullptr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c0 = (j == 0) ? 0 : p0 + 16 * (j - 1), w = (j == 0) ? p0 : min(16, p - c0), sh = (j == 0) ? M : 0;
            d4 kn = (j == 0) ? -c.k : -load_dense<false>(gw + n * M + c0, M, w, p, 0, sh, lane);
            d4 Et = (j == 0) ? c.Et : (E ? load_dense<false>(E + c0, n, w, p, 0, sh, lane) : z);
            d4 U, Xn;
            riccati_forward(KTn, kn, c.FT, c.GT, Et, X[j], U, Xn);
            X[j] = Xn;
            if (j == 0) { lqs_store<1>(rU, sMP, (unsigned)(t * M * p) * 8u, U); lqs_store(rX, sNP, (unsigned)((t + 1) * n * p) * 8u, Xn); }
            else { store_dense(uo + c0, M, w, p, 0, sh, lane, U); store_dense(xo + c0, n, w, p, 0, sh, lane, Xn); }
            if (lo) {
                d4 Wt = (j == 0) ? c.Wt : load_dense<false>(pw + n * n + c0, n, w, p, 0, sh, lane);
                d4 L = mma_tn(c.Pt, Xn, Wt);                    // P x+ + W  (P symmetric)
                if (j == 0) lqs_store(rL, sNP, (unsigned)(t * n * p) * 8u, L); else store_dense(lo + c0, n, w, p, 0, sh, lane, L);
            }
            finite = finite && tile_finite(Xn);
        }
    };
    {
        FwdTiles ta, tb;
        load_fwd(ta);
        int t = 0;
        for (; t + 1 < T; t += 2) { fstep(t, ta, tb); fstep(t + 1, tb, ta); }
        if (t < T) fstep(t, ta, tb);
    }
    int st = 0;
    if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
    if (!ok) st |= PDP_STATUS_PIVOT;
    if (lane == 0 && status) status[b] = st;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Small systems (n <= 4, m + p <= 16): FOUR trajectories per wavefront, block-diagonal in the tile (pdp_riccati_small.h).  Same inputs,
// outputs and workspace layout as lqr_solve_kernel.  A lane addresses ITS element of every operand by one 32-bit offset that is the
// same for the four trajectories (their bases differ by the batch stride: scalar registers); absent elements are zeros.
// ---------------------------------------------------------------------------------------------------------------------------------------
struct SmallOp { int offA, offB; };      // element offset in the first / second source matrix, -1 = absent
// one running pointer per trajectory of the wave (RunPtr slot r = trajectory r): the loads of a step are bare global_loads with nothing
// behind them - no bound check, select or branch - so all of them are in flight before the first is waited for
PDP_DEV RunPtr make_run_small(const pdp_mat& A, const pdp_mat& Bm, const SmallOp& o, const int* br, int t) {
    RunPtr r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.p[k] = PDP_ZERO; r.step[k] = 0;
        if (A.ptr && o.offA >= 0) { r.p[k] = mat_at(A, br[k], t) + o.offA; r.step[k] = (int)(A.tstride * 8); }
        else if (Bm.ptr && o.offB >= 0) { r.p[k] = mat_at(Bm, br[k], t) + o.offB; r.step[k] = (int)(Bm.tstride * 8); }
    }
    return r;
}
PDP_DEV double small_load(const double* __restrict__ A, const double* __restrict__ Bm, const SmallOp& o) {
    const double* q = (o.offA >= 0 && A) ? A + o.offA : ((o.offB >= 0 && Bm) ? Bm + o.offB : (const double*)PDP_ZERO);
    return *q;
}

template <int M>
__global__ void __launch_bounds__(64) lqr_solve_small_kernel(pdp_lqr_problem pr, double* __restrict__ Xo, double* __restrict__ Uo,
                                                              double* __restrict__ Lo, int32_t* __restrict__ status,
                                                              double* __restrict__ ws_gain, double* __restrict__ ws_pw) {
    static_assert(M >= 1 && M <= 4, "four rows per trajectory: m <= 4 (larger control dimensions take lqr_solve_generic_kernel)");
    const int lane = threadIdx.x, row = lane >> 4, col = lane & 15, ci = col & 3;
    const int n = pr.n, p = pr.p, T = pr.T, B = pr.B;
    const int tlane = small_transpose_lane(lane);
    const int gsz = n * M + M * p, pwsz = n * n + n * p;
    int br[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) br[r] = min(4 * (int)blockIdx.x + r, B - 1);      // the last wave may repeat the last trajectory (stores are guarded)
    const bool pc = col >= M && col < M + p;                // a parameter column of the tile
    // per-lane operand offsets (loop invariant)
    const SmallOp oRep = {(row < n && ci < n) ? row * n + ci : -1, -1};                             // n x n, rep form
    const SmallOp oRepT = {(row < n && ci < n) ? ci * n + row : -1, -1};                            // its transpose, rep form
    const SmallOp oY = {(row < n && col < M) ? row * M + col : -1, (row < n && pc) ? row * p + (col - M) : -1};     // [G | E], [Hxu | Hxe]
    const SmallOp oU = {(row < M && col < M) ? row * M + col : -1, (row < M && pc) ? row * p + (col - M) : -1};     // [Huu | Hue]
    const SmallOp oGr = {(row < n && ci < M) ? row * M + ci : -1, -1};                              // G rep
    const SmallOp oGT = {(row < M && ci < n) ? ci * M + row : -1, -1};                              // G' rep  (also Hxu' rep = Hux rep)
    const SmallOp oNP = {-1, (row < n && pc) ? row * p + (col - M) : -1};                           // n x p block behind the control columns
    const SmallOp oMP = {-1, (row < M && pc) ? row * p + (col - M) : -1};
    const SmallOp oKT = {(row < n && ci < M) ? row * M + ci : -1, -1};                              // stored K' [n][m] -> lane (k, 4b+i) = K[i][k]
    bool ok[4] = {true, true, true, true}, finite[4] = {true, true, true, true};
    d4 P, W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        P[r] = small_load(mat_at(pr.hxx, br[r], 0), nullptr, oRep);
        W[r] = small_load(nullptr, mat_at(pr.hxe, br[r], 0), oNP);
    }
    struct Bwd { d4 F, Y, Gr, Hxx, HX, HU, Hux; };
    const pdp_mat none = {nullptr, 0, 0};
    RunPtr rF = make_run_small(pr.F, none, oRep, br, T - 1), rY = make_run_small(pr.G, pr.E, oY, br, T - 1), rGr = make_run_small(pr.G, none, oGr, br, T - 1),
           rHxx = make_run_small(pr.Hxx, none, oRep, br, T - 1), rHX = make_run_small(pr.Hxu, pr.Hxe, oY, br, T - 1),
           rHU = make_run_small(pr.Huu, pr.Hue, oU, br, T - 1), rHux = make_run_small(pr.Hxu, none, oGT, br, T - 1);
    auto load_bwd = [&](Bwd& w) {       // reads the current step's operands, moves the pointers one step down
        w.F = load_run(rF, -1); w.Y = load_run(rY, -1); w.Gr = load_run(rGr, -1); w.Hxx = load_run(rHxx, -1); w.HX = load_run(rHX, -1);
        w.HU = load_run(rHU, -1); w.Hux = load_run(rHux, -1);
    };
    bool mine[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[r] = 4 * (int)blockIdx.x + r < B;
    auto bstep = [&](int t, const Bwd& c, Bwd& nx) {
        if (t > 0) load_bwd(nx);                        // the operands of step t-1 are requested before step t computes
        if (ws_pw) {                                    // P_{t+1}, W_{t+1} for the costate output (PDP.py:604)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* pw = ws_pw + ((int64_t)br[r] * T + t) * pwsz;
                if (mine[r] && oRep.offA >= 0 && col < 4) pw[oRep.offA] = P[r];
                if (mine[r] && oNP.offB >= 0) pw[n * n + oNP.offB] = W[r];
            }
        }
        // the four Riccati steps are straight-line code on separate registers: four independent MFMA chains the scheduler interleaves
        SmallGains g[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double Pr = P[r], Wr = W[r];
            ok[r] = riccati_small_backward<M>(Pr, Wr, c.F[r], c.Y[r], c.Gr[r], c.Hxx[r], c.HX[r], c.HU[r], c.Hux[r], lane, tlane, p, g[r]) && ok[r];
            P[r] = Pr; W[r] = Wr;
            finite[r] = finite[r] && fabs(Pr) <= 1.7e308 && fabs(Wr) <= 1.7e308;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* gw = ws_gain + ((int64_t)br[r] * T + t) * gsz;
            if (mine[r] && row < M && col < n) gw[col * M + row] = g[r].K;           // K' [n][m]
            if (mine[r] && oMP.offB >= 0) gw[n * M + oMP.offB] = g[r].IK;            // k [m][p]
        }
    };
    {
        Bwd ta, tb;
        load_bwd(ta);
        int t = T - 1;
        for (; t >= 1; t -= 2) { bstep(t, ta, tb); bstep(t - 1, tb, ta); }
        if (t == 0) bstep(0, ta, tb);
    }
    // ---- forward rollout (PDP.py:582-608)
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        X[r] = small_load(nullptr, mat_at(pr.X0, br[r], 0), oNP);
        if (4 * (int)blockIdx.x + r < B && oNP.offB >= 0) Xo[(int64_t)br[r] * (T + 1) * n * p + oNP.offB] = X[r];
    }
    __threadfence_block();
    struct Fwd { d4 FT, GT, KT, k, E, Pt, Wt; };
    const pdp_mat gKT = {ws_gain, (int64_t)T * gsz, gsz}, gk = {ws_gain + n * M, (int64_t)T * gsz, gsz},
                  wP = {(ws_pw && Lo) ? ws_pw : nullptr, (int64_t)T * pwsz, pwsz}, wW = {(ws_pw && Lo) ? ws_pw + n * n : nullptr, (int64_t)T * pwsz, pwsz};
    RunPtr qFT = make_run_small(pr.F, none, oRepT, br, 0), qGT = make_run_small(pr.G, none, oGT, br, 0), qKT = make_run_small(gKT, none, oKT, br, 0),
           qk = make_run_small(none, gk, oMP, br, 0), qE = make_run_small(none, pr.E, oNP, br, 0), qP = make_run_small(wP, none, oRep, br, 0),
           qW = make_run_small(none, wW, oNP, br, 0);
    auto load_fwd = [&](Fwd& w) {
        w.FT = load_run(qFT, 1); w.GT = load_run(qGT, 1); w.KT = load_run(qKT, 1); w.k = load_run(qk, 1); w.E = load_run(qE, 1);
        w.Pt = load_run(qP, 1); w.Wt = load_run(qW, 1);
    };
    auto fstep = [&](int t, const Fwd& c, Fwd& nx) {
        if (t + 1 < T) load_fwd(nx);
        double U[4], L[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double Xn;
            riccati_small_forward(-c.KT[r], -c.k[r], c.FT[r], c.GT[r], c.E[r], X[r], U[r], Xn);
            X[r] = Xn;
            L[r] = Lo ? mma4_blk(c.Pt[r], Xn, c.Wt[r]) : 0.0;                      // lambda_{t+1} = P x_{t+1} + W   (P symmetric)
            finite[r] = finite[r] && fabs(Xn) <= 1.7e308;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (mine[r] && oMP.offB >= 0) Uo[((int64_t)br[r] * T + t) * M * p + oMP.offB] = U[r];
            if (mine[r] && oNP.offB >= 0) Xo[((int64_t)br[r] * (T + 1) + t + 1) * n * p + oNP.offB] = X[r];
            if (Lo && mine[r] && oNP.offB >= 0) Lo[((int64_t)br[r] * T + t) * n * p + oNP.offB] = L[r];
        }
    };
    {
        Fwd ta, tb;
        load_fwd(ta);
        int t = 0;
        for (; t + 1 < T; t += 2) { fstep(t, ta, tb); fstep(t + 1, tb, ta); }
        if (t < T) fstep(t, ta, tb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int st = 0;
        if (!__all(finite[r])) st |= PDP_STATUS_NONFINITE;
        if (!ok[r]) st |= PDP_STATUS_PIVOT;
        if (lane == 0 && status && 4 * (int)blockIdx.x + r < B) status[br[r]] = st;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Larger systems (n > 16 or m > 4; p <= 32 per launch - more parameters go through the caller's column blocks): the reference accepts any size
// (PDP.py:446-555), so does this kernel.  Beyond one 16x16 tile per matrix the recursion runs as plain lane-parallel fp64 loops over P, W and the
// products - one wavefront per trajectory, every output element of a product owned by one lane, the m x m system solved by Gauss-Jordan with
// partial pivoting on the augmented block [Quu | Qux | Que].  The working set (3 n^2 + 2 n p + ... doubles) lives in LDS while it fits 150 KB
// (n up to ~70) and in the caller's workspace beyond (GLOBAL = true: the same code on global pointers, the lanes' hand-overs ordered by a
// workgroup-scope fence - a CU's vector cache is coherent for its own stores).  Same Schur-complement algebra, inputs, outputs and workspace
// layout as lqr_solve_kernel; not tuned - it exists so that no model size is refused (round 4: n > 32 or m > 8 returned PDP_E_SIZE).
// ---------------------------------------------------------------------------------------------------------------------------------------
constexpr int GEN_PMAX = 32;
__host__ __device__ constexpr size_t lqr_generic_lds_doubles(int n, int m, int p) {
    return (size_t)3 * n * n + (size_t)2 * n * p + (size_t)n * m + (size_t)m * (m + n + p) + (size_t)m * n + (size_t)m * p + 2 * (size_t)n * p + 8 + (size_t)m * m;
}
__host__ __device__ constexpr bool lqr_generic_in_lds(int n, int m, int p) { return lqr_generic_lds_doubles(n, m, p) * sizeof(double) <= 150 * 1024; }

template <bool GLOBAL>
__global__ void __launch_bounds__(64) lqr_solve_generic_kernel(pdp_lqr_problem pr, double* __restrict__ Xo, double* __restrict__ Uo, double* __restrict__ Lo,
                                                                int32_t* __restrict__ status, double* __restrict__ ws_gain, double* __restrict__ ws_pw,
                                                                double* __restrict__ ws_scratch) {
    extern __shared__ __attribute__((aligned(16))) double gl_lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    double* gl = GLOBAL ? ws_scratch + (int64_t)b * (int64_t)lqr_generic_lds_doubles(pr.n, pr.m, pr.p) : gl_lds;
    auto wave_lds_sync = [&]() {        // (shadows the LDS-only wait: the working set may be in global memory)
        if constexpr (GLOBAL) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const int n = pr.n, m = pr.m, p = pr.p, T = pr.T, wa = m + n + p;
    double* P = gl;                  // n x n
    double* PF = P + n * n;          // n x n
    double* Pn = PF + n * n;         // n x n
    double* W = Pn + n * n;          // n x p
    double* PEW = W + n * p;         // n x p   (P E + W)
    double* PG = PEW + n * p;        // n x m
    double* A = PG + n * m;          // m x (m + n + p): [Quu | Qux | Que] -> [I | K | k]
    double* Kk = A + m * wa;         // m x n  (forward: K), then m x p (k)
    double* kk = Kk + m * n;
    double* Xc = kk + m * p;         // n x p  (forward state)
    double* Xn = Xc + n * p;
    double* Sq = Xn + n * p + 8;     // m x m: copy of Quu for the definiteness test
    const int gsz = n * m + m * p, pwsz = n * n + n * p;
    bool ok = true, finite = true, posdef = true;
    const double* hxx = mat_at(pr.hxx, b, 0);
    const double* hxe = mat_at(pr.hxe, b, 0);
    for (int q = lane; q < n * n; q += 64) P[q] = hxx[q];
    for (int q = lane; q < n * p; q += 64) W[q] = hxe ? hxe[q] : 0.0;
    wave_lds_sync();
    for (int t = T - 1; t >= 0; --t) {
        const double *F = mat_at(pr.F, b, t), *G = mat_at(pr.G, b, t), *E = mat_at(pr.E, b, t), *Hxx = mat_at(pr.Hxx, b, t), *Hxu = mat_at(pr.Hxu, b, t),
                     *Hxe = mat_at(pr.Hxe, b, t), *Huu = mat_at(pr.Huu, b, t), *Hue = mat_at(pr.Hue, b, t);
        if (ws_pw) {
            double* pw = ws_pw + ((int64_t)b * T + t) * pwsz;
            for (int q = lane; q < n * n; q += 64) pw[q] = P[q];
            for (int q = lane; q < n * p; q += 64) pw[n * n + q] = W[q];
        }
        for (int q = lane; q < n * n; q += 64) { const int i = q / n, j = q - i * n; double s = 0.0; for (int k = 0; k < n; ++k) s += P[i * n + k] * F[k * n + j]; PF[q] = s; }
        for (int q = lane; q < n * m; q += 64) { const int i = q / m, j = q - i * m; double s = 0.0; for (int k = 0; k < n; ++k) s += P[i * n + k] * G[k * m + j]; PG[q] = s; }
        for (int q = lane; q < n * p; q += 64) { const int i = q / p, j = q - i * p; double s = W[q]; if (E) for (int k = 0; k < n; ++k) s += P[i * n + k] * E[k * p + j]; PEW[q] = s; }
        wave_lds_sync();
        for (int q = lane; q < m * wa; q += 64) {           // [Quu | Qux | Que] = [Huu | Hxu' | Hue] + G' [PG | PF | PEW]
            const int i = q / wa, c = q - i * wa;
            double s;
            if (c < m) { s = Huu[i * m + c]; for (int k = 0; k < n; ++k) s += G[k * m + i] * PG[k * m + c]; }
            else if (c < m + n) { const int j = c - m; s = Hxu ? Hxu[j * m + i] : 0.0; for (int k = 0; k < n; ++k) s += G[k * m + i] * PF[k * n + j]; }
            else { const int j = c - m - n; s = Hue ? Hue[i * p + j] : 0.0; for (int k = 0; k < n; ++k) s += G[k * m + i] * PEW[k * p + j]; }
            A[q] = s;
        }
        // Pn = Hxx + F' PF ; Wn = Hxe + F' PEW   (before the elimination overwrites nothing they need; the rank-m corrections follow)
        for (int q = lane; q < n * n; q += 64) { const int i = q / n, j = q - i * n; double s = Hxx[q]; for (int k = 0; k < n; ++k) s += F[k * n + i] * PF[k * n + j]; Pn[q] = s; }
        wave_lds_sync();
        // keep Qux' for the corrections: PG is free now -> Qux (m x n) copy in Kk ... the elimination turns A's Qux block into K
        for (int q = lane; q < m * n; q += 64) Kk[q] = A[(q / n) * wa + m + (q % n)];
        for (int q = lane; q < m * m; q += 64) Sq[q] = 0.5 * (A[(q / m) * wa + (q % m)] + A[(q % m) * wa + (q / m)]);
        wave_lds_sync();
        // Quu positive definite?  Pivots of the unpivoted elimination of its symmetric part (every lane runs the same m^3 / 3 operations on the LDS copy; the
        // writes of different lanes carry identical values)
        for (int c = 0; c < m; ++c) {
            const double d = Sq[c * m + c];
            posdef = posdef && d > 0.0;
            const double id = 1.0 / d;
            for (int r = c + 1; r < m; ++r) {
                const double f = Sq[r * m + c] * id;
                for (int j = c + 1; j < m; ++j) Sq[r * m + j] -= f * Sq[c * m + j];
            }
            wave_lds_sync();
        }
        for (int c = 0; c < m; ++c) {                       // Gauss-Jordan with partial pivoting (uniform control flow)
            int piv = c;
            double best = fabs(A[c * wa + c]);
            for (int r = c + 1; r < m; ++r) { const double v = fabs(A[r * wa + c]); if (v > best) { best = v; piv = r; } }
            if (!(best > 1e-300) || !(best <= 1.7e308)) ok = false;
            if (piv != c) { for (int q = lane; q < wa; q += 64) { const double tv = A[c * wa + q]; A[c * wa + q] = A[piv * wa + q]; A[piv * wa + q] = tv; } }
            wave_lds_sync();
            const double ip = 1.0 / A[c * wa + c];
            wave_lds_sync();
            for (int q = lane; q < wa; q += 64) A[c * wa + q] *= ip;
            wave_lds_sync();
            for (int r = 0; r < m; ++r) {
                if (r == c) continue;
                const double f = A[r * wa + c];
                wave_lds_sync();
                for (int q = lane; q < wa; q += 64) A[r * wa + q] -= f * A[c * wa + q];
                wave_lds_sync();
            }
        }
        // A = [I | K | k];  Kk holds Qux (m x n)
        double* gw = ws_gain + ((int64_t)b * T + t) * gsz;
        for (int q = lane; q < n * m; q += 64) gw[q] = A[(q % m) * wa + m + q / m];                  // K' [n][m]
        for (int q = lane; q < m * p; q += 64) gw[n * m + q] = A[(q / p) * wa + m + n + (q % p)];    // k [m][p]
        // P- = Pn - Qux' K ; W- = Hxe + F' PEW - Qux' k
        for (int q = lane; q < n * n; q += 64) { const int i = q / n, j = q - i * n; double s = Pn[q]; for (int r = 0; r < m; ++r) s -= Kk[r * n + i] * A[r * wa + m + j]; PF[q] = s; }
        for (int q = lane; q < n * p; q += 64) {
            const int i = q / p, j = q - i * p;
            double s = Hxe ? Hxe[q] : 0.0;
            for (int k = 0; k < n; ++k) s += F[k * n + i] * PEW[k * p + j];
            for (int r = 0; r < m; ++r) s -= Kk[r * n + i] * A[r * wa + m + n + j];
            W[q] = s;
            finite = finite && fabs(s) <= 1.7e308;
        }
        wave_lds_sync();
        for (int q = lane; q < n * n; q += 64) { const int i = q / n, j = q - i * n; const double s = 0.5 * (PF[q] + PF[j * n + i]); P[q] = s; finite = finite && fabs(s) <= 1.7e308; }
        wave_lds_sync();
    }
    __threadfence_block();
    // ---- forward rollout
    const double* X0 = mat_at(pr.X0, b, 0);
    for (int q = lane; q < n * p; q += 64) { const double v = X0 ? X0[q] : 0.0; Xc[q] = v; Xo[(int64_t)b * (T + 1) * n * p + q] = v; }
    wave_lds_sync();
    for (int t = 0; t < T; ++t) {
        const double *F = mat_at(pr.F, b, t), *G = mat_at(pr.G, b, t), *E = mat_at(pr.E, b, t);
        const double* gw = ws_gain + ((int64_t)b * T + t) * gsz;
        double* uo = Uo + ((int64_t)b * T + t) * m * p;
        for (int q = lane; q < m * p; q += 64) {            // U = -K X - k
            const int i = q / p, j = q - i * p;
            double s = -gw[n * m + q];
            for (int k = 0; k < n; ++k) s -= gw[k * m + i] * Xc[k * p + j];
            kk[q] = s; uo[q] = s;
        }
        wave_lds_sync();
        double* xo = Xo + ((int64_t)b * (T + 1) + t + 1) * n * p;
        for (int q = lane; q < n * p; q += 64) {            // X+ = F X + G U + E
            const int i = q / p, j = q - i * p;
            double s = E ? E[q] : 0.0;
            for (int k = 0; k < n; ++k) s += F[i * n + k] * Xc[k * p + j];
            for (int r = 0; r < m; ++r) s += G[i * m + r] * kk[r * p + j];
            Xn[q] = s; xo[q] = s;
            finite = finite && fabs(s) <= 1.7e308;
        }
        wave_lds_sync();
        if (Lo) {
            const double* pw = ws_pw + ((int64_t)b * T + t) * pwsz;
            double* lo = Lo + ((int64_t)b * T + t) * n * p;
            for (int q = lane; q < n * p; q += 64) { const int i = q / p, j = q - i * p; double s = pw[n * n + q]; for (int k = 0; k < n; ++k) s += pw[i * n + k] * Xn[k * p + j]; lo[q] = s; }
        }
        for (int q = lane; q < n * p; q += 64) Xc[q] = Xn[q];
        wave_lds_sync();
    }
    int st = 0;
    if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
    if (!ok) st |= PDP_STATUS_PIVOT;
    if (!posdef) st |= PDP_STATUS_INDEFINITE;
    if (lane == 0 && status) status[b] = st;
}

}  // namespace pdp
