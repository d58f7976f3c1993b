// pdp_policy.h - the two policy parameterisations of ControlPlanning (reference PDP/PDP.py:699-759) as device code.
//   POLY : Lagrange polynomial in time, u(t) = sum_i b_i(t) U_i, theta = vcat(U_0..U_N), b_i(t) = prod_{j!=i} (t - tau_j)/(tau_i - tau_j)
//          (PDP.py:705-716; factors applied in the same left-to-right order), d pi/dx = 0, d pi/d theta = [b_0 I_m ... b_N I_m].
//   MLP  : a = A_0 x + b_0 ; a = A_k tanh(a) + b_k ; theta = [vec_F(A_0), b_0, vec_F(A_1), b_1, ...] column-major (PDP.py:733-751).
//   TABLE: u(t) = sum_i table[t][i] U_i with the basis values given per step (the warped / recovery-matrix variants, PDP.py:882-1141), otherwise as POLY.
#pragma once
#include "../../include/pdp_hip.h"
#include "pdp_tile.h"

namespace pdp {

constexpr int MLP_MAX_WIDTH = 32;

PDP_DEV double lagrange_basis(const pdp_policy& pol, int i, double t) {
    double b = 1.0;
    for (int j = 0; j < pol.n_pivots; ++j)
        if (j != i) b = b * (t - pol.pivots[j]) / (pol.pivots[i] - pol.pivots[j]);
    return b;
}

// u = pi(t, x, theta).  act (optional, MLP): pre-activations a_k of every layer, act[k*MLP_MAX_WIDTH + row]
template <int NX, int NU>
PDP_DEV void policy_eval(const pdp_policy& pol, int t, const double* x, const double* __restrict__ theta, double* u, double* act = nullptr) {
    if (pol.kind == PDP_POLICY_POLY || pol.kind == PDP_POLICY_TABLE) {
        const bool tab = pol.kind == PDP_POLICY_TABLE;
        const int nb = tab ? pol.n_basis : pol.n_pivots;
#pragma unroll
        for (int j = 0; j < NU; ++j) u[j] = 0.0;
        for (int i = 0; i < nb; ++i) {
            double b = tab ? pol.table[(int64_t)t * nb + i] : lagrange_basis(pol, i, (double)t);
#pragma unroll
            for (int j = 0; j < NU; ++j) u[j] += b * theta[i * NU + j];
        }
        return;
    }
    constexpr int WM = NX > MLP_MAX_WIDTH ? NX : MLP_MAX_WIDTH;            // (a model with more states than the widest layer: the input sets the size)
    double z[WM], a[WM];
    int cols = NX, off = 0;
    for (int i = 0; i < NX; ++i) z[i] = x[i];
    for (int k = 0; k < pol.n_layers; ++k) {
        const int rows = pol.sizes[k];
        for (int r = 0; r < rows; ++r) {
            double s = 0.0;
            for (int c = 0; c < cols; ++c) s += theta[off + r + c * rows] * z[c];      // column-major A_k
            a[r] = s + theta[off + rows * cols + r];
        }
        if (act) for (int r = 0; r < rows; ++r) act[k * MLP_MAX_WIDTH + r] = a[r];
        off += rows * cols + rows;
        if (k + 1 < pol.n_layers) for (int r = 0; r < rows; ++r) z[r] = pdp_tanh(a[r]);
        cols = rows;
    }
    for (int j = 0; j < NU; ++j) u[j] = a[j];
}

// d pi/dx [NU x NX] and d pi/d theta [NU x p], row-major, at (t, x).
template <int NX, int NU>
PDP_DEV void policy_jacobians(const pdp_policy& pol, int p, int t, const double* x, const double* __restrict__ theta, double* dUx, double* dUe) {
    if (pol.kind == PDP_POLICY_POLY || pol.kind == PDP_POLICY_TABLE) {
        const bool tab = pol.kind == PDP_POLICY_TABLE;
        const int nb = tab ? pol.n_basis : pol.n_pivots;
        for (int i = 0; i < NU * NX; ++i) dUx[i] = 0.0;
        for (int i = 0; i < NU * p; ++i) dUe[i] = 0.0;
        for (int i = 0; i < nb; ++i) {
            double b = tab ? pol.table[(int64_t)t * nb + i] : lagrange_basis(pol, i, (double)t);
            for (int j = 0; j < NU; ++j) dUe[j * p + i * NU + j] = b;
        }
        return;
    }
    double act[8 * MLP_MAX_WIDTH], u[NU];
    policy_eval<NX, NU>(pol, t, x, theta, u, act);
    // layer offsets
    int offs[8], colsk[8];
    {
        int cols = NX, off = 0;
        for (int k = 0; k < pol.n_layers; ++k) { offs[k] = off; colsk[k] = cols; off += pol.sizes[k] * cols + pol.sizes[k]; cols = pol.sizes[k]; }
    }
    // J = d u / d a_k (NU x rows_k), start with identity at the output layer
    constexpr int WM = NX > MLP_MAX_WIDTH ? NX : MLP_MAX_WIDTH;
    double J[NU * WM], Jz[NU * WM];
    for (int j = 0; j < NU; ++j) for (int r = 0; r < NU; ++r) J[j * WM + r] = (j == r) ? 1.0 : 0.0;
    for (int k = pol.n_layers - 1; k >= 0; --k) {
        const int rows = pol.sizes[k], cols = colsk[k], off = offs[k];
        // input of layer k: z = x (k = 0) or tanh(a_{k-1})
        for (int j = 0; j < NU; ++j) {
            for (int c = 0; c < cols; ++c) {
                double zc = (k == 0) ? x[c] : pdp_tanh(act[(k - 1) * MLP_MAX_WIDTH + c]);
                for (int r = 0; r < rows; ++r) dUe[j * p + off + r + c * rows] = J[j * WM + r] * zc;
            }
            for (int r = 0; r < rows; ++r) dUe[j * p + off + rows * cols + r] = J[j * WM + r];
        }
        for (int j = 0; j < NU; ++j)
            for (int c = 0; c < cols; ++c) {
                double s = 0.0;
                for (int r = 0; r < rows; ++r) s += J[j * WM + r] * theta[off + r + c * rows];
                Jz[j * WM + c] = s;
            }
        if (k > 0) {
            for (int j = 0; j < NU; ++j)
                for (int c = 0; c < cols; ++c) { double th_ = pdp_tanh(act[(k - 1) * MLP_MAX_WIDTH + c]); J[j * WM + c] = Jz[j * WM + c] * (1.0 - th_ * th_); }
        } else {
            for (int j = 0; j < NU; ++j) for (int c = 0; c < NX; ++c) dUx[j * NX + c] = Jz[j * WM + c];
        }
    }
}

}  // namespace pdp
