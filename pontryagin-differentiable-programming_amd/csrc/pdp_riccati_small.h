// pdp_riccati_small.h - the auxiliary-control-system solve (LQR.lqrSolver, reference PDP/PDP.py:557-608) for SMALL systems, n <= 4
// (pendulum, cart-pole, robot arm), FOUR trajectories per wavefront.
//
// A 16x16 tile holds a block-diagonal batch: tile register r (rows 4r .. 4r+3 of the tile) belongs to trajectory r of the wave's four,
// and every product of the recursion becomes ONE v_mfma_f64_4x4x4_4b per trajectory - the 4-block MFMA multiplies a 4x4 left factor
// into a 4x16 right factor (25-32 cycles) where the padded one-trajectory-per-tile form spends 4 x 64 cycles on a 16x16x16 product
// that is 94 % zeros for n = 4.  The four trajectories are four independent dependency chains in one instruction stream.
//
// "R4" layout of one trajectory's 4 x 16 matrix: one double per lane, element (row, col) in lane 16 row + col.
// "rep" form of a 4 x 4 matrix A: A replicated in the four column blocks, lane (k, 4 b + i) holds A[k][i] - the left-operand form:
//       mma4_blk(rep(A), Y, C)[i][c] = C[i][c] + sum_k A[k][i] Y[k][c] = (C + A' Y)[i][c]                       (pdp_tile.h)
// and a right operand in rep form gives a result in rep form.  Per step (Schur form, same algebra as pdp_riccati.h):
//     PF  = P F                 PY2 = P [G|E] + [0|W]          FY = [Hxu|Hxe] + F' PY2 = [Qux' | Wn]
//     Q2  = [Huu|Hue] + G' PY2 = [Quu | Que]                   Qux = Hux + G' PF           Pn = Hxx + F' PF
//     K = Quu^-1 Qux            [I|k] = Quu^-1 Q2               P- = Pn - Qux' K            W- = Wn - Qux' k
// 10 small MFMAs per trajectory and step; forward step U = -K X - k, X+ = F X + G U + E: 3.
#pragma once
#include "pdp_riccati.h"

namespace pdp {

// transpose inside every 4 x 4 column block of an R4 value: lane (k, 4b + i) <-> lane (i, 4b + k)
PDP_DEV int small_transpose_lane(int lane) { return 16 * (lane & 3) + (lane & 12) + (lane >> 4); }
PDP_DEV double small_transpose(double v, int tlane) { return __shfl(v, tlane, 64); }

struct SmallGains {
    double K;    // rows < m: K = Quu^-1 Qux in rep form (lane (i, 4b + k) = K[i][k])
    double IK;   // rows < m: [I | k], k in columns m .. m+p0-1
    bool pd;     // Quu positive definite (the multiple-shooting solver's inertia test)
};

// One backward step for ONE trajectory (one register of each packed tile).  Prep: P in rep form (symmetric), W2: [0 | W].
// Frep, Hxxrep: rep form; Y2 = [G | E], HX2 = [Hxu | Hxe], HU2 = [Huu | Hue] (rows < m): R4; Grep: lane (k, 4b + i) = G[k][i] (i < m);
// Huxrep: lane (i, 4b + k) = Hxu[k][i] (i < m).  Returns false on a vanishing / non-finite pivot of the m x m system.
// TSYM: how P- is symmetrised.  false: (P- + shuffle-transposed P-)/2 - two ds_bpermute round trips at the end of the step's dependency chain.  true: the transpose is
// COMPUTED beside P-, (Pn - Qux'K)' = (Hxx + (PF)'F) - K'Qux, two more small MFMAs that run in the shadow of the ones they mirror - the same products summed in the same
// order, so with a bitwise symmetric Hxx (generated Hessians: the two halves share one pool slot) the result is bit-identical to the shuffle's; callers whose
// Hxx is user data (lqr_solve_small_kernel) keep the shuffle, which is symmetric whatever it is given.
template <int M, bool TSYM = false>
PDP_DEV bool riccati_small_backward(double& Prep, double& W2, double Frep, double Y2, double Grep, double Hxxrep, double HX2, double HU2, double Huxrep,
                                    int lane, int tlane, int p0, SmallGains& g) {
    static_assert(M >= 1 && M <= 4, "the small-system algebra keeps the m x m control block in rows 0..3 of a tile: m <= 4");
    const int row = lane >> 4, col = lane & 15;
    const double PF = mma4_blk(Prep, Frep, 0.0);          // (P F) rep                       (P symmetric)
    const double PY2 = mma4_blk(Prep, Y2, W2);            // [P G | P E + W]
    const double FY = mma4_blk(Frep, PY2, HX2);           // [Qux' | Wn]
    const double Q2 = mma4_blk(Grep, PY2, HU2);           // rows < m: [Quu | Que]
    const double Qux = mma4_blk(Grep, PF, Huxrep);        // rows < m: Qux, rep
    const double Pn = mma4_blk(Frep, PF, Hxxrep);         // (Hxx + F' P F) rep
    double a[M * M], ai[M * M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) a[i * M + j] = readlane_f64(Q2, 16 * i + j);
    bool ok;
    if constexpr (M == 4) {
        ok = inverse_small<4>(a, ai);
        // positive definite iff every pivot of the unpivoted elimination is positive (uniform values)
        double l[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) l[q] = a[q];
        bool pos = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pos = pos && l[k * 4 + k] > 0.0;
            const double ip = 1.0 / l[k * 4 + k];
#pragma unroll
            for (int i = k + 1; i < 4; ++i) {
                const double f = l[i * 4 + k] * ip;
#pragma unroll
                for (int j = k + 1; j < 4; ++j) l[i * 4 + j] -= f * l[k * 4 + j];
            }
        }
        g.pd = pos;
    } else {
        ok = inverse_small_fast<M>(a, ai);
        g.pd = posdef_small<M>(a);
    }
    double Zrep = 0.0;                                    // lane (k, 4b + i) = Quu^-1[i][k]
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) Zrep = (row == i && (col & 3) == j) ? ai[j * M + i] : Zrep;
    g.K = mma4_blk(Zrep, Qux, 0.0);                       // rows < m: K rep
    double IK = mma4_blk(Zrep, Q2, 0.0);                  // rows < m: [I | k]
    double Pm = mma4_blk(-Qux, g.K, Pn);                  // Pn - Qux' K    (rep)
    const double Wn = mma4_blk(-Qux, IK, FY);             // [~0 | Wn - Qux' k]
    const bool pcol = col >= M && col < M + p0;
    W2 = pcol ? Wn : 0.0;                                 // the control columns are zero only up to rounding: masked (see pdp_riccati.h)
    g.IK = pcol ? IK : 0.0;
    // P <- (P + P')/2: the skew rounding error would be amplified step by step
    if constexpr (TSYM) {
        const double PnT = mma4_blk(PF, Frep, Hxxrep);    // Hxx + (P F)' F = Pn'
        const double PmT = mma4_blk(-g.K, Qux, PnT);      // Pn' - K' Qux = (Pn - Qux' K)'
        Prep = 0.5 * (Pm + PmT);
    } else Prep = 0.5 * (Pm + small_transpose(Pm, tlane));
    return ok;
}

// Forward step of one trajectory: U = -K X - k (rows < m), X+ = F X + G U + E.  KTrepneg: lane (k, 4b + i) = -K[i][k];
// kneg: rows < m: -k in columns m..; FTrep: lane (k, 4b + i) = F[i][k]; GTrep: lane (k, 4b + i) = G[i][k] (k < m).
PDP_DEV void riccati_small_forward(double KTrepneg, double kneg, double FTrep, double GTrep, double E2, double X, double& U, double& Xn) {
    U = mma4_blk(KTrepneg, X, kneg);
    Xn = mma4_blk(FTrep, X, E2);
    Xn = mma4_blk(GTrep, U, Xn);
}

}  // namespace pdp
