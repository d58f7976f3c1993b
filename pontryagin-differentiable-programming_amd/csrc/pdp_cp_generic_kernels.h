// pdp_cp_generic_kernels.h - ControlPlanning.step (reference PDP/PDP.py:850-878) for EVERYTHING the tuned kernels refuse, and the home of the warped / recovery-matrix
// variants (PDP.py:882-1141): cp_step_generic_kernel.
//
// The reference has no size limits (PDP.py:727-759: any list of hidden layers; 699-725: any number of pivots; 1081-1114: one parameter per control and time step).  The
// tuned kernels have: cp_step_mlp16_kernel <= 4 layers x 16 units, cp_step_adjoint_kernel <= 8 x 32 and p <= 512, the tile kernels n <= 16, m <= 4, p <= 64.  Round 4
// returned PDP_E_SIZE beyond that ("a user with hidden [64, 64] does not fit", verdict).  This kernel takes whatever is left - correctness first:
//   * one wavefront per trajectory, the same adjoint formulation as cp_step_adjoint_kernel (rollout; mu_T = h_x; v_t = c_u + G_t' mu_{t+1};
//     grad += (d pi / d theta)' v_t; mu_t = c_x + F_t' mu_{t+1} + (d pi / d x)' v_t), O(T (n^2 + p)) instead of the reference's O(T n^2 p);
//   * every loop over a layer's rows, the parameters or the states is strided over the 64 lanes - no width, layer count (<= 16: the policy struct), parameter count or
//     state dimension (<= 64 per wavefront pass of the Jacobian columns: static_assert) is assumed;
//   * the parameters are read where they are (global memory, L2-resident), the trajectory, the controls and the hidden activations of every step live in a caller-owned
//     workspace, the gradient is accumulated in its output row (read-modify-write by the lane that owns the entry); only the current step's layer inputs / deltas and
//     the pool of Jacobian rows are in LDS - and move to the workspace too when a network is too wide for that (pointers of either address space, agent-scope fences).
//
// PDP_POLICY_TABLE (new):  u_t = sum_i table[t][i] theta[i m .. i m + m)  with a dense [T][n_basis] table in device memory.  That is
//   * recmat_* (PDP.py:1081-1141): table[t][i] = 1 if step t lies in grid cell i - theta IS the control of every cell, the gradient is H_u summed per cell: the whole
//     "recovery matrix" step as ONE launch (round 4: rollout, costates, getAuxSys and an index_add, with a host hop);
//   * warp_*  (PDP.py:960-1008): table[t][i] = b_i(cell(t)), the Lagrange basis on the cell index - computed once at warp_init_step, not per step.
#pragma once
#include "pdp_model_kernels.h"

namespace pdp {

constexpr int GEN_MAXL = 16;       // = the length of pdp_policy.sizes

// what lives where.  Offsets in doubles; *_g: offset inside the trajectory's workspace slice (global), *_l: inside LDS; in_lds says which one zs / ds use.
struct CpGenLayout {
    int64_t ws_per_traj;           // doubles of workspace per trajectory
    int64_t x_g, u_g, act_g, zs_g, ds_g, vs_g;
    int zs_l, ds_l, mu_l, v_l, ul_l, tab_l, th_l, pool_l, lds_total;
    int sum_in, sum_w, actw, rows, wide, ul, tab, thl;     // sum_in: NX + hidden widths (+ 2 constants); sum_w: all widths; wide: zs / ds in the workspace; ul: u_t / v_t of an open-loop policy staged in LDS
};

template <class Mdl>
// wide_bytes: above this many bytes of layer inputs + deltas the two arrays live in the workspace instead of LDS (96 KB; PDP_CP_GENERIC_WIDE_BYTES overrides it on the host so
// that a test can send a SMALL network down that route and compare it bit for bit with the LDS route)
__host__ __device__ inline CpGenLayout cp_generic_layout(const pdp_policy& pol, int T, bool have_x, bool have_u, int wide_bytes = 96 * 1024) {
    constexpr int STRIDE = (Mdl::PATH_NVAR + 1 + Mdl::PATH_NCONST) | 1;
    CpGenLayout L;
    int sum_in = Mdl::NX, sum_w = 0, actw = 0;
    if (pol.kind == PDP_POLICY_MLP) {
        for (int k = 0; k < pol.n_layers; ++k) { sum_w += pol.sizes[k]; if (k + 1 < pol.n_layers) { sum_in += pol.sizes[k]; actw += pol.sizes[k]; } }
    } else {
        const int nb = pol.kind == PDP_POLICY_POLY ? pol.n_pivots : pol.n_basis;
        sum_in = nb; sum_w = Mdl::NU;
    }
    L.sum_in = sum_in + 2; L.sum_w = sum_w; L.actw = actw;
    L.wide = (L.sum_in + L.sum_w) * 8 > wide_bytes;
    int o = 0;
    L.zs_l = o; o += L.wide ? 0 : L.sum_in;
    L.ds_l = o; o += L.wide ? 0 : L.sum_w;
    L.mu_l = o; o += Mdl::NX;
    L.v_l = o; o += Mdl::NU + Mdl::NX;
    // open-loop policies (POLY, TABLE): all controls are computed before the rollout and read from LDS inside it (a global load in the serial chain costs a memory round
    // trip per step); the same area then holds v_t = c_u + G_t' mu_{t+1} of every step, from which the gradient is formed after the sweep
    L.ul = pol.kind != PDP_POLICY_MLP && (int64_t)T * Mdl::NU <= 4096;
    L.ul_l = o; o += L.ul ? T * Mdl::NU : 0;
    // a table policy's basis values [T][n_basis] in LDS while they fit 48 KB: the control pre-pass and the final gradient contraction walk the table with dependent
    // accumulations - from global memory every trip waits for its load (measured on the recmat drivers: 60 of 74 us per launch)
    L.tab = pol.kind == PDP_POLICY_TABLE && (int64_t)T * pol.n_basis <= 6144;
    L.tab_l = o; o += L.tab ? T * pol.n_basis : 0;
    const int p_open = (pol.kind == PDP_POLICY_POLY ? pol.n_pivots : pol.n_basis) * Mdl::NU;
    L.thl = pol.kind != PDP_POLICY_MLP && p_open <= 2048;      // ... and the parameters of an open-loop policy beside it
    L.th_l = o; o += L.thl ? p_open : 0;
    L.pool_l = o;
    int rows = (int)((150 * 1024 / 8 - o - 8) / STRIDE);
    rows = rows > 64 ? 64 : rows;
    // keep the footprint near 40 KB (four wavefronts per CU) when that still leaves 8 rows
    const int rows40 = (40 * 1024 / 8 - o - 8) / STRIDE;
    if (rows40 >= 8) rows = rows40 > 64 ? 64 : rows40;
    L.rows = rows;
    L.lds_total = o + (rows > 0 ? rows : 0) * STRIDE + 8;
    int64_t g = 0;
    L.x_g = g; g += have_x ? 0 : (int64_t)(T + 1) * Mdl::NX;
    L.u_g = g; g += have_u ? 0 : (int64_t)T * Mdl::NU;
    L.act_g = g; g += (int64_t)T * actw;
    L.zs_g = g; g += L.wide ? L.sum_in : 0;
    L.ds_g = g; g += L.wide ? L.sum_w : 0;
    L.vs_g = g; g += (pol.kind != PDP_POLICY_MLP && !L.ul) ? (int64_t)T * Mdl::NU : 0;
    L.ws_per_traj = g;
    return L;
}

// MLPK: the policy kind as a template constant - the open-loop instantiation (Lagrange, table: what recmat / warp run) carries none of the layer tables, whose 96 scalar
// registers spilled into VGPR lanes and cost the recmat drivers' single-trajectory step a third of its time
template <class Mdl, bool MLPK>
__global__ void __launch_bounds__(64) cp_step_generic_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0, const double* __restrict__ theta, int tb,
                                                              double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ xo, double* __restrict__ uo,
                                                              double* __restrict__ ws, CpGenLayout L) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    constexpr int NV = Mdl::PATH_NVAR, STRIDE = (NV + 1 + Mdl::PATH_NCONST) | 1;      // pool row: [entries | 0.0 | constants]
    // (one lane per Jacobian column: NX, NU <= 64 - checked by the launcher, which returns PDP_E_SIZE beyond that; a static_assert here would make a 65-state model
    //  fail to COMPILE its whole library)
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    double* wsb = ws + (int64_t)b * L.ws_per_traj;
    double* xs = xo ? xo + (int64_t)b * (T + 1) * NX : wsb + L.x_g;                   // the API outputs double as the staging of the trajectory
    double* us = uo ? uo + (int64_t)b * T * NU : wsb + L.u_g;
    double* acts = wsb + L.act_g;                                                     // [T][actw]: tanh outputs of the hidden layers
    // Arrays that live in LDS or - when they do not fit - in the workspace are reached through accessors that branch (uniformly) on the flag: a pointer that may be either
    // is a GENERIC pointer to the compiler, every access a flat_load / flat_store through the texture path - in the serial chains of the rollout and the sweep that was
    // most of a recmat step.  Zr / Zw: layer inputs of the current step (z_0 = x | z_1 | ... | 1.0 | 0.0); Dr / Dw: layer deltas.
    auto Zr = [&](int i) -> double { return L.wide ? wsb[L.zs_g + i] : lds[L.zs_l + i]; };
    auto Zw = [&](int i, double v) { if (L.wide) wsb[L.zs_g + i] = v; else lds[L.zs_l + i] = v; };
    auto Dr = [&](int i) -> double { return L.wide ? wsb[L.ds_g + i] : lds[L.ds_l + i]; };
    auto Dw = [&](int i, double v) { if (L.wide) wsb[L.ds_g + i] = v; else lds[L.ds_l + i] = v; };
    double *mu = lds + L.mu_l, *vv = lds + L.v_l, *pool = lds + L.pool_l;
    const double* thb = theta + (int64_t)b * tb;
    double* gb = grad + (int64_t)b * p;
    constexpr bool mlp = MLPK;
    const bool table = pol.kind == PDP_POLICY_TABLE;
    const int nl = mlp ? pol.n_layers : 0, nb = mlp ? 0 : (table ? pol.n_basis : pol.n_pivots);
    const int one = L.sum_in - 2;                                                     // layer-input slots `one`, `one + 1` hold 1.0 (bias factor) and 0.0
    auto sync = [&]() {       // everything the lanes exchange goes through LDS or (wide networks, trajectory, activations, gradient) global memory of this wavefront's own slice
        // (workgroup scope: the wavefront is its own workgroup and a CU's vector cache is coherent for its own stores - the fence is the wait for them)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    };
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    // layer tables (uniform): parameter offset, rows, cols, offset of the layer's input in zs, of its delta in ds, of its stored activation
    constexpr int NLT = MLPK ? GEN_MAXL : 1;
    int loff[NLT], lrows[NLT], lcols[NLT], zoff[NLT], doff[NLT], aoff[NLT];
    {
        int cols = NX, off = 0, zo = 0, dof = 0, ao = 0;
        for (int k = 0; k < NLT; ++k) {
            loff[k] = off; lcols[k] = cols; lrows[k] = (k < nl) ? pol.sizes[k] : 0; zoff[k] = zo; doff[k] = dof; aoff[k] = ao;
            if (k < nl) { off += lrows[k] * cols + lrows[k]; zo += cols; dof += lrows[k]; if (k + 1 < nl) ao += lrows[k]; cols = lrows[k]; }
        }
    }
    for (int j = lane; j < p; j += 64) gb[j] = 0.0;
    if (lane == 0) { Zw(one, 1.0); Zw(one + 1, 0.0); }
    sync();

    // open-loop policies: u_t = sum_i b_i(t) theta_i for ALL t at once (lane = step), into the trajectory's control array and the LDS staging
    auto TAB = [&](int64_t i) -> double { return L.tab ? lds[L.tab_l + i] : pol.table[i]; };
    auto THO = [&](int i) -> double { return L.thl ? lds[L.th_l + i] : thb[i]; };      // parameters as the control pre-pass reads them
    if (L.tab) for (int q = lane; q < T * nb; q += 64) lds[L.tab_l + q] = pol.table[q];
    if (L.thl) for (int q = lane; q < p; q += 64) lds[L.th_l + q] = thb[q];
    if (L.tab || L.thl) sync();
    auto basis = [&](int t, int i) { return table ? TAB((int64_t)t * nb + i) : lagrange_basis(pol, i, (double)t); };
    auto ULr = [&](int i) -> double { return L.ul ? lds[L.ul_l + i] : us[i]; };          // u_t inside the rollout ...
    auto VSr = [&](int i) -> double { return L.ul ? lds[L.ul_l + i] : wsb[L.vs_g + i]; };  // ... v_t inside the sweep (the same LDS area)
    auto VSw = [&](int i, double v) { if (L.ul) lds[L.ul_l + i] = v; else wsb[L.vs_g + i] = v; };
    if (!mlp) {
        for (int t = lane; t < T; t += 64) {
            for (int j = 0; j < NU; ++j) {
                double u = 0.0;
                for (int i = 0; i < nb; ++i) u += basis(t, i) * THO(i * NU + j);       // i ascending, as policy_eval
                us[(int64_t)t * NU + j] = u;
                if (L.ul) lds[L.ul_l + t * NU + j] = u;
            }
        }
        sync();
    }
    // MLP: u_t = pi(x_t, theta) into vv[0 .. NU); layer inputs stay in zs, hidden activations go to acts[t]
    auto policy_forward = [&](int t) {
        for (int i = lane; i < NX; i += 64) Zw(i, xs[(int64_t)t * NX + i]);
        sync();
        for (int k = 0; k < nl; ++k) {
            const int rows = lrows[k], cols = lcols[k], zo = zoff[k];
            for (int r = lane; r < rows; r += 64) {
                double a = 0.0;
                for (int c = 0; c < cols; ++c) a += thb[loff[k] + r + c * rows] * Zr(zo + c);    // column-major A_k (PDP.py:739), c ascending as policy_eval
                a += thb[loff[k] + rows * cols + r];
                if (k + 1 < nl) { const double z = pdp_tanh(a); Zw(zoff[k + 1] + r, z); acts[(int64_t)t * L.actw + aoff[k] + r] = z; }
                else vv[r] = a;
            }
            sync();
        }
    };

    // ---------------- forward rollout (PDP.py:763-786), executed uniformly by the wavefront
    double J = 0.0;
    {
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        for (int i = lane; i < NX; i += 64) xs[i] = x0[(int64_t)b * NX + i];
        sync();
        for (int t = 0; t < T; ++t) {
            if (mlp) {
                policy_forward(t);
#pragma unroll
                for (int j = 0; j < NU; ++j) uc[j] = vv[j];
                for (int j = lane; j < NU; j += 64) us[(int64_t)t * NU + j] = vv[j];
            } else {
#pragma unroll
                for (int j = 0; j < NU; ++j) uc[j] = ULr(t * NU + j);
            }
            Mdl::dyn(xc, uc, nullptr, pc, xn);
            J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[(int64_t)(t + 1) * NX + i] = xn[i];
            }
            if (mlp) sync();                           // (the next step's policy reads x_{t+1} back; an open-loop rollout has nothing to wait for)
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        double h[NX];
        Mdl::dhx(xc, nullptr, pc, h);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) mu[i] = h[i];
        }
        sync();
    }

    // ---------------- adjoint sweep
    // per-lane pool slots: column `lane` of F (lane < NX) and of G (lane < NU), entries c_x[lane], c_u[lane]; a slot is an entry, the row's 0.0 or one of its constants
    int fo[NX], go[NX], cxo, cuo;
    auto enc = [&](int code) { return code >= 0 ? code : (code == -1 ? NV : NV + 1 + (-2 - code)); };
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        fo[k] = enc(lane < NX ? Mdl::path_code(0, k * NX + lane) : -1);
        go[k] = enc(lane < NU ? Mdl::path_code(1, k * NU + lane) : -1);
    }
    cxo = enc(lane < NX ? Mdl::path_code(2, lane) : -1);
    cuo = enc(lane < NU ? Mdl::path_code(3, lane) : -1);
    const int CH = L.rows;
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int t0 = c * ch, cnt = min(ch, T - t0);
        sync();
        if (lane < cnt) {                               // lane = time step: F, G, c_x, c_u at (x_t, u_t) of the stored trajectory
            const int t = t0 + lane;
            double xc[NX], uc[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xs[(int64_t)t * NX + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) uc[j] = us[(int64_t)t * NU + j];
            double* row = pool + lane * STRIDE;
            PackedSink sk{row};
            Mdl::eval_path(xc, uc, nullptr, nullptr, pc, sk);
            row[NV] = 0.0;
#pragma unroll
            for (int i = 0; i < Mdl::PATH_NCONST; ++i) row[NV + 1 + i] = Mdl::path_const(i);
        }
        sync();
        for (int tl = cnt - 1; tl >= 0; --tl) {
            const int t = t0 + tl;
            const double* rowt = pool + tl * STRIDE;
            // v = c_u + G' mu
            if (lane < NU) {
                double a = rowt[cuo];
#pragma unroll
                for (int k = 0; k < NX; ++k) a += rowt[go[k]] * mu[k];
                vv[lane] = a;
            }
            for (int i = lane; i < NX; i += 64) vv[NU + i] = 0.0;          // (d pi/dx)' v: stays 0 for the open-loop policies
            sync();
            if (!mlp) {
                for (int j = lane; j < NU; j += 64) VSw(t * NU + j, vv[j]);           // the gradient of an open-loop policy is formed after the sweep
            } else {
                for (int i = lane; i < NX; i += 64) Zw(i, xs[(int64_t)t * NX + i]);
                for (int k = 1; k < nl; ++k) for (int r = lane; r < lrows[k - 1]; r += 64) Zw(zoff[k] + r, acts[(int64_t)t * L.actw + aoff[k - 1] + r]);
                for (int j = lane; j < NU; j += 64) Dw(doff[nl - 1] + j, vv[j]);
                sync();
                for (int k = nl - 1; k >= 0; --k) {
                    const int rows = lrows[k], cols = lcols[k];
                    const int dko = doff[k];
                    // parameters of layer k: vec_F(A_k)[r + c rows] gets delta_k[r] z_k[c], the bias delta_k[r]
                    {
                        const int nw = rows * cols;
                        int e = lane, r = lane % rows, cc = lane / rows;                 // e = r + cc rows, advanced by 64 per trip without divisions
                        const int dr = 64 % rows, dc = 64 / rows;
                        for (; e < nw; e += 64) {
                            gb[loff[k] + e] += Dr(dko + r) * Zr(zoff[k] + cc);
                            r += dr; cc += dc;
                            if (r >= rows) { r -= rows; ++cc; }
                        }
                        for (int r2 = lane; r2 < rows; r2 += 64) gb[loff[k] + nw + r2] += Dr(dko + r2);
                    }
                    // back through the layer: (A_k' delta_k)[c], times tanh' of the layer below, or (d pi/dx)' v at the input
                    for (int cidx = lane; cidx < cols; cidx += 64) {
                        double a = 0.0;
                        for (int r = 0; r < rows; ++r) a += thb[loff[k] + r + cidx * rows] * Dr(dko + r);
                        if (k > 0) { const double zk = Zr(zoff[k] + cidx); Dw(doff[k - 1] + cidx, a * (1.0 - zk * zk)); }
                        else vv[NU + cidx] = a;
                    }
                    sync();
                }
            }
            // mu_t = c_x + F' mu_{t+1} + (d pi/dx)' v
            double m_new = 0.0;
            if (lane < NX) {
                m_new = rowt[cxo] + vv[NU + lane];
#pragma unroll
                for (int k = 0; k < NX; ++k) m_new += rowt[fo[k]] * mu[k];
            }
            sync();
            if (lane < NX) mu[lane] = m_new;
            sync();
        }
    }
    if (!mlp) {       // theta = vcat(U_0 .. U_N):  d cost / d theta[i m + j] = sum_t b_i(t) v_t[j]
        sync();
        for (int q = lane; q < p; q += 64) {
            const int i = q / NU, j = q - i * NU;
            double g = 0.0;
            for (int t = 0; t < T; ++t) g += basis(t, i) * VSr(t * NU + j);
            gb[q] = g;
        }
    }
    if (lane == 0) loss[b] = J;
}

// d pi / d x [m][n] and d pi / d theta [m][p] of a tanh-MLP of any shape at every (trajectory, time step) - the policy half of ControlPlanning.getAuxSys
// (PDP.py:788-811, dpolicy_dx_fn / dpolicy_de_fn of 754-759) for networks beyond cp_auxsys_kernel's lane-local arrays (8 layers x 32 units).  One wavefront per (b, t):
// forward pass with the layer inputs kept in LDS, then one backward pass per output row j (delta_out = e_j): d u_j / d vec_F(A_k)[r + c rows] = delta_k[r] z_k[c],
// d u_j / d b_k[r] = delta_k[r], delta_{k-1} = (A_k' delta_k) (1 - z_k^2), d u_j / d x = A_0' delta_0.  LDS: sum of the layer inputs + two delta vectors.
template <int NX, int NU>
__global__ void __launch_bounds__(64) cp_policy_jac_generic_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x, const double* __restrict__ theta, int tb,
                                                                    double* __restrict__ dUx, double* __restrict__ dUe, int sum_in, int maxw) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int64_t bt = blockIdx.x;
    const int b = (int)(bt / T), t = (int)(bt - (int64_t)b * T), lane = threadIdx.x;
    double *zs = lds, *da = lds + sum_in, *db = da + maxw;
    const double* thb = theta + (int64_t)b * tb;
    const double* xc = x + ((int64_t)b * (T + 1) + t) * NX;
    const int nl = pol.n_layers;
    int loff[GEN_MAXL], lrows[GEN_MAXL], lcols[GEN_MAXL], zoff[GEN_MAXL];
    {
        int cols = NX, off = 0, zo = 0;
        for (int k = 0; k < GEN_MAXL; ++k) {
            loff[k] = off; lcols[k] = cols; lrows[k] = (k < nl) ? pol.sizes[k] : 0; zoff[k] = zo;
            if (k < nl) { off += lrows[k] * cols + lrows[k]; zo += cols; cols = lrows[k]; }
        }
    }
    for (int i = lane; i < NX; i += 64) zs[i] = xc[i];
    wave_lds_sync();
    for (int k = 0; k + 1 < nl; ++k) {
        const int rows = lrows[k], cols = lcols[k];
        for (int r = lane; r < rows; r += 64) {
            double a = 0.0;
            for (int c = 0; c < cols; ++c) a += thb[loff[k] + r + c * rows] * zs[zoff[k] + c];
            zs[zoff[k + 1] + r] = pdp_tanh(a + thb[loff[k] + rows * cols + r]);
        }
        wave_lds_sync();
    }
    double* ue = dUe + bt * NU * p;
    double* ux = dUx + bt * NU * NX;
    for (int j = 0; j < NU; ++j) {
        double* dk = da;
        double* dn = db;
        for (int r = lane; r < lrows[nl - 1]; r += 64) dk[r] = r == j ? 1.0 : 0.0;
        wave_lds_sync();
        for (int k = nl - 1; k >= 0; --k) {
            const int rows = lrows[k], cols = lcols[k], nw = rows * cols;
            for (int e = lane; e < nw; e += 64) { const int c = e / rows, r = e - c * rows; ue[(int64_t)j * p + loff[k] + e] = dk[r] * zs[zoff[k] + c]; }
            for (int r = lane; r < rows; r += 64) ue[(int64_t)j * p + loff[k] + nw + r] = dk[r];
            for (int c = lane; c < cols; c += 64) {
                double a = 0.0;
                for (int r = 0; r < rows; ++r) a += thb[loff[k] + r + c * rows] * dk[r];
                if (k > 0) { const double z = zs[zoff[k] + c]; dn[c] = a * (1.0 - z * z); }
                else ux[j * NX + c] = a;
            }
            wave_lds_sync();
            double* tmp = dk; dk = dn; dn = tmp;
        }
    }
}

}  // namespace pdp
