// pdp_riccati.h - one time step of the auxiliary-control-system solve (LQR.lqrSolver, reference
// PDP/PDP.py:557-608) on register tiles, one wavefront per trajectory.
//
// Mathematics.  The reference writes the backward step with (I + P R)^-1 and Huu^-1 (PDP.py:563-580):
//     A = F - G Huu^-1 Hxu',  R = G Huu^-1 G',  M = E - G Huu^-1 Hue,  Q = Hxx - Hxu Huu^-1 Hxu',
//     N = Hxe - Hxu Huu^-1 Hue,  S = A'(I + P R)^-1,  P- = Q + S P A,  W- = N + S (W + P M)
// By the matrix-inversion lemma this is the Schur complement of the control block of
//     Theta = [F G E]' P [F G E] + [Hxx Hxu Hxe; Hxu' Huu Hue; . . .]  (+ [0 0 F'W; 0 0 G'W]):
//     Quu = Huu + G'PG,  Qux = Hxu' + G'PF,  Que = Hue + G'(PE + W)
//     K = Quu^-1 Qux,  k = Quu^-1 Que,  P- = Hxx + F'PF - Qux'K,  W- = Hxe + F'(PE + W) - Qux'k
// and the forward pass (PDP.py:588-608)  u = -K x - k,  x+ = F x + G u + E,  lambda+ = P x+ + W.
// Only an m x m system is solved per step (m <= 4) instead of two n x n inversions; results agree with the
// reference order of operations to rounding (tests: <= 1e-10 relative against oracle/pdp_oracle.py).
//
// Tile packing (16 columns per tile): the n x p sensitivity block shares a tile with the n x m control
// block - Y2 = [G | E], HX2 = [Hxu | Hxe], HU2 = [Huu | Hue], W2 = [0 | W] - so one 16x16x16 product yields
// both [PG | PE+W], one yields [Qux' | W-part], one yields [Quu | Que].  Columns m .. m+p0-1 carry the first
// p0 = min(p, 16-m) parameters; further parameters live in extra tiles (E_j, Hxe_j, Hue_j, W_j), 16 each.
// Backward step: 5 full products + 5 rank-m products = 25 MFMAs (1600 cycles) for p <= 16 - m.
#pragma once
#include "pdp_tile.h"

namespace pdp {

constexpr int RICCATI_SCRATCH = 272 + 272 + 64;   // HX2 (stride 17, HUX_FROM_HX2 only) | P (stride 17) | rows 0..3 of Q2

PDP_DEV void tile_to_lds17(double* s, const d4 v, int lane) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[tile_row(lane, r) * 17 + tile_col(lane)] = v[r];
}
PDP_DEV d4 tile_from_lds17_transposed(const double* s, int lane) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = s[tile_col(lane) * 17 + tile_row(lane, r)];
    return v;
}

struct RiccatiGains {
    d4 KT;   // K^T  (n x m) in columns 0..m-1               -> forward: U = -(KT)^T X - k   (only if WANT_KT)
    d4 K;    // K = Quu^-1 Qux (m x n), rows 0..m-1
    d4 IK;   // rows 0..m-1: [ I | k_0 ] (k_0 in columns m..m+p0-1)
    d4 Z;    // Quu^-T (m x m, top-left)
    double Zrep;   // the same block replicated in the four column blocks (register 0): operand of the 4-row products
    d4 Qux;  // m x n, rows 0..m-1
    bool pd; // WANT_PD only: Quu is positive definite (all leading principal minors > 0) - the inertia test of the KKT system
};

// Sylvester's criterion on a row-major M x M matrix held uniformly by the wave (M <= 3; the M = 4 path has its own form)
template <int M>
PDP_DEV bool posdef_small(const double* a) {
    if constexpr (M == 1) return a[0] > 0.0;
    else if constexpr (M == 2) return a[0] > 0.0 && a[0] * a[3] - a[1] * a[2] > 0.0;
    else {
        const double d2 = a[0] * a[4] - a[1] * a[3];
        const double d3 = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
        return a[0] > 0.0 && d2 > 0.0 && d3 > 0.0;
    }
}

// Symmetry.  P is symmetric in exact arithmetic and the products use P^T in place of P.  Left alone, the
// skew-symmetric rounding error of P- = Hxx + F'PF - Qux'K is amplified from step to step (measured on the
// reference's quadrotor demo, T=50: 3.4e-10 relative error in X vs a 50-digit solution, against 1e-15 once P
// is re-symmetrised each step), so every step ends with P <- (P + P^T)/2 through a padded LDS transpose.
//
// scratch: LDS, RICCATI_SCRATCH doubles, private to the wave (one wave per workgroup).  Returns false when the
// m x m solve meets a vanishing / non-finite pivot.  No barrier and no global-memory wait inside.
// Grep: G (n x m, m <= 4) replicated in the four column blocks of a tile, the operand form of the 4-row products (mma4_tn).
// WANT_KT: also form K^T as a tile (one more 16x16x4 MFMA) - callers that store the gains as K^T [n][m]; the fused kernel stores K
// itself and reads it back transposed, which costs nothing.
// Hux0: Hxu^T (m x n) as a "rows 0..3" register.  Qux = Hux + G' (P F) is formed directly as a 4-row product - in exact arithmetic
// the transpose of the first m columns of FY, which used to be taken through an LDS round trip in the middle of the step.
// HUX_FROM_HX2: Hux is taken from the Hxu block of HX2 by an LDS transpose issued at the top of the step (off the critical path: it
// does not depend on P) instead of being passed in - for callers that have no transposed copy of Hxu at hand.
// WANT_PD: g.pd reports whether Quu is positive definite (multiple-shooting OC solver: the KKT matrix of the Newton step has the
// right inertia iff every Quu of the sweep is positive definite).
#ifndef PDP_RB_T
#define PDP_RB_T(i)      // timing builds (probes/phase_timing3.py) define it to take a cycle stamp
#endif
template <int M, bool WANT_KT = true, bool HUX_FROM_HX2 = false, bool WANT_PD = false>
PDP_DEV bool riccati_backward(d4& P, d4& W0, const d4 Ft, const d4 Y2, const d4 Grep, const d4 Hxx, const d4 HX2, const d4 HU2, double Hux0,
                              double* scratch, int lane, int p0, RiccatiGains& g, d4& P_old_out) {
    const d4 z = zero4();
    if constexpr (HUX_FROM_HX2) tile_to_lds17(scratch, HX2, lane);
    PDP_RB_T(0);
    d4 PF = mma_tn(P, Ft, z);        // P F        (P symmetric)
    d4 PY2 = mma_tn(P, Y2, W0);      // [P G | P E + W]
    PDP_RB_T(1);
    d4 FY = mma_tn(Ft, PY2, HX2);    // [Hxu + F'PG | Hxe + F'(PE+W)] = [Qux' | Wn]
    d4 Q2 = z;
    Q2[0] = mma4_tn(Grep, PY2, HU2[0]);   // [Quu | Que] = [Huu | Hue] + G' [PG | PE+W]: 4 rows, 4 small MFMAs
    PDP_RB_T(2);
    if constexpr (HUX_FROM_HX2) {
        wave_lds_sync();
        Hux0 = ((lane >> 4) < M) ? scratch[(lane & 15) * 17 + (lane >> 4)] : 0.0;      // Hux[i][c] = Hxu[c][i]
    }
    d4 Qux = z;
    Qux[0] = mma4_tn(Grep, PF, Hux0);     // Qux = Hux + G'PF   (m x n)
    d4 Pn = mma_tn(Ft, PF, Hxx);     // Hxx + F'PF
    P_old_out = P;
    PDP_RB_T(3);
    // ---- m x m system Quu (element (i,j) lives in lane 16 i + j, register 0)
    const int row = lane >> 4, col = lane & 15;
    d4 Z = z;                                   // Z = Quu^-T in the top-left corner
    double Zrep = 0.0;                          // ... and in every 4-column block
    bool ok = true;
    if constexpr (M == 4) {
        // lane-parallel cofactor inverse: lane (i,j) = (row, col) computes the cofactor C_ij of its own element from the
        // 3x3 minor read out of LDS (9 reads at loop-invariant per-lane addresses); det = sum_c a_0c C_0c via v_readlane;
        // Z[i][j] = inv[j][i] = C_ij / det.  ~27 fp64 operations per lane instead of ~140 for the uniform adjugate.
        scratch[544 + lane] = Q2[0];            // rows 0..3 of Q2, flat [row*16 + col]
        wave_lds_sync();
        PDP_RB_T(4);
        const int i = row, j = col & 3;
        const int r0 = (i == 0) ? 1 : 0, r1 = (i <= 1) ? 2 : 1, r2 = (i <= 2) ? 3 : 2;
        const int c0 = (j == 0) ? 1 : 0, c1 = (j <= 1) ? 2 : 1, c2 = (j <= 2) ? 3 : 2;
        const double* q = scratch + 544;
        const double m00 = q[r0 * 16 + c0], m01 = q[r0 * 16 + c1], m02 = q[r0 * 16 + c2];
        const double m10 = q[r1 * 16 + c0], m11 = q[r1 * 16 + c1], m12 = q[r1 * 16 + c2];
        const double m20 = q[r2 * 16 + c0], m21 = q[r2 * 16 + c1], m22 = q[r2 * 16 + c2];
        double cof = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
        cof = ((i + j) & 1) ? -cof : cof;
        // Laplace expansion along row 0; the size of its terms against the size of their sum is the conditioning guard
        const double ac = Q2[0] * cof;               // lanes 0..3 hold a_0c C_0c (their element is (0, c)): one product, then 4 broadcasts
        const double t0 = readlane_f64(ac, 0), t1 = readlane_f64(ac, 1), t2 = readlane_f64(ac, 2), t3 = readlane_f64(ac, 3);
        const double det = (t0 + t1) + (t2 + t3), mag = (fabs(t0) + fabs(t1)) + (fabs(t2) + fabs(t3));
        if constexpr (WANT_PD) {      // leading minors: a00, a00 a11 - a01 a10, cofactor C33 (lane 51 = element (3,3)), det
            const double a00 = readlane_f64(Q2[0], 0), a01 = readlane_f64(Q2[0], 1), a10 = readlane_f64(Q2[0], 16), a11 = readlane_f64(Q2[0], 17);
            g.pd = a00 > 0.0 && a00 * a11 - a01 * a10 > 0.0 && readlane_f64(cof, 51) > 0.0 && det > 0.0;
        }
        if (fabs(det) > 1e-10 * mag && fabs(det) <= 1.7e308) {      // uniform branch
            // 1/det: hardware reciprocal + one Newton step (the full IEEE division sequence is a 12-instruction dependent chain
            // in front of the gain MFMAs; det is nowhere near the subnormal / overflow ranges that sequence exists for)
            double rdet = __builtin_amdgcn_rcp(det);
            rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
            Zrep = cof * rdet;                   // (the cofactor was computed for j = col & 3: already replicated)
        } else {                                 // ill-conditioned / singular: pivoted Gauss-Jordan, uniform over the wave
            double a[16], ai[16];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a[ii * 4 + jj] = readlane_f64(Q2[0], 16 * ii + jj);
            ok = inverse_small<4>(a, ai);
            double zz = 0.0;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) zz = (row == ii && (col & 3) == jj) ? ai[jj * 4 + ii] : zz;
            Zrep = zz;
        }
    } else {
        double a[M * M], ai[M * M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) a[i * M + j] = readlane_f64(Q2[0], 16 * i + j);
        ok = inverse_small_fast<M>(a, ai);
        if constexpr (WANT_PD) g.pd = posdef_small<M>(a);
        double zz = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) zz = (row == i && (col & 3) == j) ? ai[j * M + i] : zz;
        Zrep = zz;
    }
    Z[0] = (col < 4) ? Zrep : 0.0;
    PDP_RB_T(5);
    d4 K = z;
    K[0] = mma4_blk(Zrep, Qux[0], 0.0);   // Quu^-1 Qux            (m x n)
    g.IK = z;
    g.IK[0] = mma4_blk(Zrep, Q2[0], 0.0); // Quu^-1 [Quu | Que] = [I | k]
    if constexpr (WANT_KT) g.KT = mma_tn_r0(Qux, Z, z);          // Qux' Quu^-T = K'      (n x m)
    g.K = K;
    g.Z = Z;
    g.Zrep = Zrep;
    g.Qux = Qux;
    P = mms_tn_r0(Qux, K, Pn);            // Hxx + F'PF - Qux'K
    tile_to_lds17(scratch + 272, P, lane);
    PDP_RB_T(6);
    d4 Wn = mms_tn_r0(Qux, g.IK, FY);     // [Qux' - Qux' I | Wn - Qux' k]
    // columns < M of Wn are Qux' - Qux' (Z Quu): zero only up to the rounding of Z Quu ~ I.  Left in, they perturb P G of the next
    // step and the error compounds over the horizon (quadrotor T = 50: parity lost) - they are masked out.
    W0 = keep_cols(Wn, M, M + p0, lane);
    g.IK = keep_cols(g.IK, M, M + p0, lane);
    wave_lds_sync();
    PDP_RB_T(7);
    P = 0.5 * (P + tile_from_lds17_transposed(scratch + 272, lane));
    PDP_RB_T(8);
    return ok;
}

// Homogeneous form of the backward step for a problem with ONE affine column (the Newton step of the multiple-shooting OC solver): the state is augmented
// by a constant 1, x~ = [dx; 1] (n + 1 <= 16), so that the affine terms ride inside the tiles -
//     F~ = [F c; 0 1]   G~ = [G; 0]   Hxx~ = [Hxx rx; rx' 0]   Hux~ = [Hxu' | ru]   P~ = [P W; W' s]
// and the step is the plain Riccati step of a purely quadratic problem (same algebra as riccati_backward, K~ = [K | k], P~- = Hxx~ + F~'P~F~ - Qux~'K~,
// whose (x, 1) block is W- = rx + F'(P c + W) - Qux' k): no separate W recursion, no [G | c] / [Hxu | rx] / [Huu | ru] packing, 13 full-tile and 9
// four-row MFMAs instead of 18 and 10.  The (1, 1) element of P~ carries the constant of the cost-to-go (never used).
// Ft = F~ tile, Grep = G~ replicated in the four column blocks, HU0 = Huu (rows < M, columns < M of the FIRST column block only), Hux0 = Hux~ (rows < M, n + 1 columns).
// Quu = Huu + G'(P G) without a full-tile product (round 6; P~ G~ used to be one: 4 x 64 cycles for 4 useful columns, and 4 more four-row MFMAs on top):
//     PGb = sum_q mfma4(P[q], Grep[q])      lane 16 i + 4 b + j  <-  (P~ G~)[4 b + i][j]      (P~ symmetric: the row block b of P~ G~ lands in COLUMN block b)
//     Quu_b = mfma4(Gblk, PGb)              lane 16 i + 4 b + j  <-  sum_k G[4 b + k][i] (P G)[4 b + k][j]      Gblk: lane 16 k + 4 b + i  <-  G~[4 b + k][i]
//     Quu = Huu + sum_b Quu_b               two rotations inside the rows of 16 lanes (DPP row_ror 8, 4): every column block ends with the whole sum
// 5 four-row MFMAs and 2 adds instead of 4 full-tile and 4 four-row MFMAs: ~180 cycles of a 2075-cycle stage.
template <int CTRL>
PDP_DEV double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int M, bool WANT_PD = true>
PDP_DEV bool riccati_backward_aug(d4& P, const d4 Ft, const double Gblk, const d4 Grep, const d4 Hxx, const double HU0, const double Hux0, double* scratch, int lane,
                                  RiccatiGains& g) {
    const d4 z = zero4();
    d4 PF = mma_tn(P, Ft, z);             // P~ F~      (P~ symmetric)
    const double PGb = mma4_tn(P, Grep, 0.0);      // (P~ G~) by row blocks (see above)
    d4 Q2 = z;
    {
        double q = mma4_blk(Gblk, PGb, HU0);       // Huu (column block 0 only) + the four partial sums of G'(P G)
        q += dpp_mov_f64<0x128>(q);                // row_ror:8
        q += dpp_mov_f64<0x124>(q);                // row_ror:4
        Q2[0] = q;                                 // Quu, replicated in the four column blocks
    }
    d4 Qux = z;
    Qux[0] = mma4_tn(Grep, PF, Hux0);     // Qux~ = Hux~ + G~'P~F~   (m x (n + 1)): [Qux | Que]
    d4 Pn = mma_tn(Ft, PF, Hxx);          // Hxx~ + F~'P~F~
    const int row = lane >> 4, col = lane & 15;
    double Zrep = 0.0;                    // Quu^-T replicated in every 4-column block
    bool ok = true;
    if constexpr (M == 4) {               // lane-parallel cofactor inverse (see riccati_backward)
        scratch[544 + lane] = Q2[0];
        wave_lds_sync();
        const int i = row, j = col & 3;
        const int r0 = (i == 0) ? 1 : 0, r1 = (i <= 1) ? 2 : 1, r2 = (i <= 2) ? 3 : 2;
        const int c0 = (j == 0) ? 1 : 0, c1 = (j <= 1) ? 2 : 1, c2 = (j <= 2) ? 3 : 2;
        const double* q = scratch + 544;
        const double m00 = q[r0 * 16 + c0], m01 = q[r0 * 16 + c1], m02 = q[r0 * 16 + c2];
        const double m10 = q[r1 * 16 + c0], m11 = q[r1 * 16 + c1], m12 = q[r1 * 16 + c2];
        const double m20 = q[r2 * 16 + c0], m21 = q[r2 * 16 + c1], m22 = q[r2 * 16 + c2];
        double cof = m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
        cof = ((i + j) & 1) ? -cof : cof;
        const double ac = Q2[0] * cof;
        const double t0 = readlane_f64(ac, 0), t1 = readlane_f64(ac, 1), t2 = readlane_f64(ac, 2), t3 = readlane_f64(ac, 3);
        const double det = (t0 + t1) + (t2 + t3), mag = (fabs(t0) + fabs(t1)) + (fabs(t2) + fabs(t3));
        if constexpr (WANT_PD) {
            const double a00 = readlane_f64(Q2[0], 0), a01 = readlane_f64(Q2[0], 1), a10 = readlane_f64(Q2[0], 16), a11 = readlane_f64(Q2[0], 17);
            g.pd = a00 > 0.0 && a00 * a11 - a01 * a10 > 0.0 && readlane_f64(cof, 51) > 0.0 && det > 0.0;
        }
        if (fabs(det) > 1e-10 * mag && fabs(det) <= 1.7e308) {
            double rdet = __builtin_amdgcn_rcp(det);
            rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
            Zrep = cof * rdet;
        } else {
            double a[16], ai[16];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a[ii * 4 + jj] = readlane_f64(Q2[0], 16 * ii + jj);
            ok = inverse_small<4>(a, ai);
            double zz = 0.0;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) zz = (row == ii && (col & 3) == jj) ? ai[jj * 4 + ii] : zz;
            Zrep = zz;
        }
    } else {
        double a[M * M], ai[M * M];
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) a[i * M + j] = readlane_f64(Q2[0], 16 * i + j);
        ok = inverse_small_fast<M>(a, ai);
        if constexpr (WANT_PD) g.pd = posdef_small<M>(a);
        double zz = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) zz = (row == i && (col & 3) == j) ? ai[j * M + i] : zz;
        Zrep = zz;
    }
    d4 K = z;
    K[0] = mma4_blk(Zrep, Qux[0], 0.0);   // [K | k] = Quu^-1 Qux~
    g.K = K;
    g.Zrep = Zrep;
    g.Qux = Qux;
    P = mms_tn_r0(Qux, K, Pn);            // Hxx~ + F~'P~F~ - Qux~'K~
    tile_to_lds17(scratch + 272, P, lane);
    wave_lds_sync();
    P = 0.5 * (P + tile_from_lds17_transposed(scratch + 272, lane));
    return ok;
}

// Extra parameter tile j >= 1 (16 columns of E / Hxe / Hue / W, unshifted).  P_old = P before the update.
PDP_DEV void riccati_backward_extra(const d4 P_old, d4& Wj, const d4 Ft, const d4 Grep, const d4 Ej, const d4 Hxej, const d4 Huej,
                                    const RiccatiGains& g, d4& kj) {
    d4 S1 = mma_tn(P_old, Ej, Wj);        // P E_j + W_j
    d4 Wn = mma_tn(Ft, S1, Hxej);         // Hxe_j + F'(..)
    const double Que = mma4_tn(Grep, S1, Huej[0]);      // Hue_j + G'(..)  (m rows)
    kj = zero4();
    kj[0] = mma4_blk(g.Zrep, Que, 0.0);   // Quu^-1 Que_j
    Wj = mms_tn_r0(g.Qux, kj, Wn);
}

// Forward step for one parameter tile: U = -K X - k ; X+ = F X + G U + E.   FT = F^T tile, GT = G^T tile (m x n).
// KTrepneg: -K^T (n x m) replicated in the four column blocks (operand form of the 4-row product).
PDP_DEV void riccati_forward(const d4 KTrepneg, const d4 kneg, const d4 FT, const d4 GT, const d4 Etile, const d4 X, d4& U, d4& Xn) {
    U = zero4();
    U[0] = mma4_tn(KTrepneg, X, kneg[0]); // -(K X) - k   (m rows, 4 small MFMAs)
    Xn = mma_tn(FT, X, Etile);            // F X + E
    Xn = mma_tn_r0(GT, U, Xn);            // + G U
}

}  // namespace pdp
