// pdp_model_kernels.h - batched kernels instantiated once per generated model (struct PdpModel, see
// codegen.py).  Section B of include/pdp_hip.h.
//
// Work decomposition
//   * serial-in-time scalar recursions (rollout, costates) : one LANE per trajectory (materialised API), or
//     executed uniformly by the wavefront that owns the trajectory (fused kernel);
//   * model derivative evaluation (the aux system of OCSys.getAuxSys, reference PDP/PDP.py:287-301) is
//     independent across time steps -> one LANE per time step; in the fused kernel a chunk of CHUNK steps is
//     evaluated at once and only the structurally non-zero entries are written, packed, to an LDS pool;
//   * the Riccati / sensitivity recursions run on MFMA register tiles (pdp_riccati.h), which gather their
//     operands from the LDS pool through per-lane offsets computed once per kernel.
// The per-step Jacobians/Hessians therefore never touch HBM in the fused path; HBM sees x0, u, theta, the demo,
// the state/costate trajectories (API outputs), the feedback gains (scratch) and loss/gradient.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/pdp_hip.h"
#include "pdp_riccati.h"
#include "pdp_riccati_small.h"
#include "pdp_policy.h"

namespace pdp {

// ------------------------------------------------------------------------------------------------------
// sinks for the generated eval_<group>() functions
// ------------------------------------------------------------------------------------------------------
struct PackedSink {            // variable entry k -> base[k]   (LDS pool row of this lane's time step)
    double* base;
    template <int K> PDP_DEV void put(double v) { base[K] = v; }
};

template <class Mdl> struct PathDense {     // scatter into the dense API layout: group 'path' of CP / SYSID models
    double* p[4];
    template <int K> PDP_DEV void put(double v) {
        constexpr int mat = Mdl::PATH_MAT[K], off = Mdl::PATH_OFF[K];
        if (p[mat]) p[mat][off] = v;
    }
};

template <class Mdl>
PDP_DEV void load_theta(const double* __restrict__ theta, int b, int bstride, double* th) {
#pragma unroll
    for (int k = 0; k < (Mdl::NP > 0 ? Mdl::NP : 1); ++k) th[k] = (Mdl::NP > 0) ? theta[(int64_t)b * bstride + k] : 0.0;
}

// ------------------------------------------------------------------------------------------------------
// OC: rollout / costate (lane per trajectory), aux system (lane per (b,t))
// ------------------------------------------------------------------------------------------------------
template <class Mdl>
__global__ void __launch_bounds__(64) oc_rollout_kernel(int B, int T, const double* __restrict__ x0, const double* __restrict__ u, const double* __restrict__ theta,
                                  int tb, double* __restrict__ x, double* __restrict__ cost) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    double xc[NX], xn[NX], uc[NU];
    double* xb = x + (int64_t)b * (T + 1) * NX;
#pragma unroll
    for (int i = 0; i < NX; ++i) { xc[i] = x0[(int64_t)b * NX + i]; xb[i] = xc[i]; }
    double J = 0.0;
    // a lane walks its own trajectory: every per-step load is an uncoalesced HBM round trip, so the operands of step t+1 are
    // requested before step t computes (the addresses do not depend on the recursion)
    const double* ub = u + (int64_t)b * T * NU;
    double un[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) un[i] = ub[i];
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
        for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = ub[tn * NU + i]; }
        Mdl::dyn(xc, uc, th, pc, xn);
        if (cost) J += Mdl::path_cost(xc, uc, th, pc);
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = xn[i]; xb[(t + 1) * NX + i] = xn[i]; }
    }
    if (cost) cost[b] = J + Mdl::final_cost(xc, th, pc);
}

// closed-loop rollout u = ubar - alpha k - K (x - xbar) of ONE lane; gains[t] = {K^T [n][m], k [m]}.  The reference trajectory and the
// gains of step t+1 are requested before step t computes (uncoalesced per-lane loads: an HBM round trip each otherwise).
template <class Mdl>
PDP_DEV double closed_loop_rollout(int T, double a, const double* __restrict__ x0, const double* __restrict__ ub, const double* __restrict__ xr,
                                   const double* __restrict__ gb, const double* th, const double* pc, double* __restrict__ xo, double* __restrict__ uo) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, GSZ = NX * NU + NU;
    double xc[NX], xn[NX], uc[NU], gn[GSZ], un[NU], xrn[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xc[i] = x0[i]; xo[i] = xc[i]; xrn[i] = xr[i]; }
#pragma unroll
    for (int i = 0; i < GSZ; ++i) gn[i] = gb[i];
#pragma unroll
    for (int i = 0; i < NU; ++i) un[i] = ub[i];
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
        double g[GSZ], ur[NU], xref[NX];
#pragma unroll
        for (int i = 0; i < GSZ; ++i) { g[i] = gn[i]; gn[i] = gb[tn * GSZ + i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { ur[i] = un[i]; un[i] = ub[tn * NU + i]; }
#pragma unroll
        for (int i = 0; i < NX; ++i) { xref[i] = xrn[i]; xrn[i] = xr[tn * NX + i]; }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            double v = ur[j] - a * g[NX * NU + j];
#pragma unroll
            for (int i = 0; i < NX; ++i) v -= g[i * NU + j] * (xc[i] - xref[i]);
            uc[j] = v;
            uo[t * NU + j] = v;
        }
        Mdl::dyn(xc, uc, th, pc, xn);
        J += Mdl::path_cost(xc, uc, th, pc);
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = xn[i]; xo[(t + 1) * NX + i] = xn[i]; }
    }
    return J + Mdl::final_cost(xc, th, pc);
}

// closed-loop rollout (one lane per trajectory), per-sample step length alpha
template <class Mdl>
__global__ void __launch_bounds__(64) oc_rollout_feedback_kernel(int B, int T, const double* __restrict__ x0, const double* __restrict__ ubar, const double* __restrict__ xbar,
                                           const double* __restrict__ gains, const double* __restrict__ alpha, const double* __restrict__ theta, int tb,
                                           double* __restrict__ x, double* __restrict__ u, double* __restrict__ cost) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, GSZ = NX * NU + NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    cost[b] = closed_loop_rollout<Mdl>(T, alpha[b], x0 + (int64_t)b * NX, ubar + (int64_t)b * T * NU, xbar + (int64_t)b * (T + 1) * NX,
                                       gains + (int64_t)b * T * GSZ, th, pc, x + (int64_t)b * (T + 1) * NX, u + (int64_t)b * T * NU);
}

template <class Mdl>
__global__ void __launch_bounds__(64) oc_costate_kernel(int B, int T, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ theta,
                                  int tb, double* __restrict__ lam) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    const double* xb = x + (int64_t)b * (T + 1) * NX;
    const double* ub = u + (int64_t)b * T * NU;
    double* lb = lam + (int64_t)b * T * NX;
    double xc[NX], uc[NU], lc[NX], ln[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = xb[T * NX + i];
    Mdl::dhx(xc, th, pc, lc);                                  // lam[T-1] = h_x(x_T)
#pragma unroll
    for (int i = 0; i < NX; ++i) lb[(T - 1) * NX + i] = lc[i];
    double xq[NX], uq[NU];                                 // (x_k, u_k) of the next step, requested one step ahead
#pragma unroll
    for (int i = 0; i < NX; ++i) xq[i] = xb[(T - 1) * NX + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) uq[i] = ub[(T - 1) * NU + i];
    for (int k = T - 1; k >= 1; --k) {                     // lam[k-1] = c_x(x_k,u_k) + f_x' lam[k]
        const int kn = k > 1 ? k - 1 : 1;
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = xq[i]; xq[i] = xb[kn * NX + i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) { uc[i] = uq[i]; uq[i] = ub[kn * NU + i]; }
        Mdl::costate_step(xc, uc, lc, th, pc, ln);
#pragma unroll
        for (int i = 0; i < NX; ++i) { lc[i] = ln[i]; lb[(k - 1) * NX + i] = ln[i]; }
    }
}

// Residuals of the multiple-shooting NLP of OCSys.ocSolver (reference PDP/PDP.py:131-182) at a point (x, u, lam), lam[t] = multiplier of f(x_t, u_t) - x_{t+1}:
//     c [B][T][n]      defects f(x_t, u_t) - x_{t+1}
//     rx [B][T+1][n]   grad_x of the Lagrangian: node 0: 0 (x_0 is fixed), node 0 < t < T: H_x(x_t, u_t, lam_t) - lam_{t-1}, node T: h_x(x_T) - lam_{T-1}
//     ru [B][T][m]     H_u(x_t, u_t, lam_t)
//     cost [B][T+1]    path cost of stage t, final cost at T
// One lane per (trajectory, node): size-generic (any n, m the model has), the building block of the kernel-by-kernel multiple-shooting route for problems
// beyond the solver kernels' tiles (ocsolver.solve_batch_ms_generic); same quantities as the trial pass of oc_solve_ms2_kernel.
template <class Mdl>
__global__ void __launch_bounds__(64) oc_ms_residuals_kernel(int B, int T, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ lam,
                                       const double* __restrict__ theta, int tb, double* __restrict__ c, double* __restrict__ rx, double* __restrict__ ru,
                                       double* __restrict__ cost) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * (T + 1)) return;
    const int b = (int)(idx / (T + 1)), t = (int)(idx - (int64_t)b * (T + 1));
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    const double* xb = x + ((int64_t)b * (T + 1) + t) * NX;
    double xc[NX], v[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = xb[i];
    double* rxb = rx + ((int64_t)b * (T + 1) + t) * NX;
    if (t < T) {
        const double* ub = u + ((int64_t)b * T + t) * NU;
        const double* lb = lam + ((int64_t)b * T + t) * NX;
        double uc[NU], lc[NX], hu[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) uc[i] = ub[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) lc[i] = lb[i];
        Mdl::dyn(xc, uc, th, pc, v);
        double* cb = c + ((int64_t)b * T + t) * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) cb[i] = v[i] - xb[NX + i];
        Mdl::costate_step(xc, uc, lc, th, pc, v);               // c_x + f_x' lam_t = H_x
#pragma unroll
        for (int i = 0; i < NX; ++i) rxb[i] = t > 0 ? v[i] - lb[i - NX] : 0.0;
        Mdl::dHu(xc, uc, lc, th, pc, hu);
        double* rub = ru + ((int64_t)b * T + t) * NU;
#pragma unroll
        for (int i = 0; i < NU; ++i) rub[i] = hu[i];
        cost[(int64_t)b * (T + 1) + t] = Mdl::path_cost(xc, uc, th, pc);
    } else {
        Mdl::dhx(xc, th, pc, v);
        const double* lb = lam + ((int64_t)b * T + (T - 1)) * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) rxb[i] = v[i] - lb[i];
        cost[(int64_t)b * (T + 1) + T] = Mdl::final_cost(xc, th, pc);
    }
}

// OCSys.getAuxSys, materialised.  One wavefront per (trajectory, chunk of CHUNK time steps): lane = time step evaluates the
// generated code into the packed LDS pool (as in the fused kernel), then the wave expands every matrix family into the dense
// API layout [B][T][rows][cols] with COALESCED stores - the chunk's slice of a family is one contiguous run of cnt*rows*cols
// doubles.  (A lane-per-(b,t) scatter of 5.8 KB per thread ran at 1.3 TB/s of 8-byte stores; this form is bound by the
// 314 KB per trajectory it has to write.)
// steps per workgroup of the materialised getAuxSys kernel: write-bound, wants many small workgroups in flight
template <class Mdl>
__host__ __device__ constexpr int auxsys_chunk() { return Mdl::CHUNK < 16 ? Mdl::CHUNK : 16; }

template <class Mdl, int MAT>
PDP_DEV void auxsys_expand(const double* blk, const short* codes, int nc, int stride, int cnt, double* __restrict__ dst, int lane,
                           double diag = 0.0) {      // diag: added to the diagonal (Levenberg-Marquardt damping of Huu in the OC solver)
    constexpr int RC = Mdl::PATH_ROWS[MAT] * Mdl::PATH_COLS[MAT];
    if (!dst || RC == 0) return;
    for (int q = lane; q < cnt * RC; q += 64) {
        const int tl = q / RC, i = q - tl * RC;
        const int code = codes[i];
        const double v = code >= 0 ? blk[nc + tl * stride + code] : blk[code == -1 ? 0 : 1 + (-2 - code)];
        dst[q] = (diag != 0.0 && i % (Mdl::PATH_COLS[MAT] + 1) == 0) ? v + diag : v;
    }
}

template <class Mdl>
__global__ void __launch_bounds__(64) oc_auxsys_kernel(int B, int T, const double* __restrict__ x, const double* __restrict__ u,
                                                        const double* __restrict__ lam, const double* __restrict__ theta, int tb, pdp_oc_auxsys o) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = auxsys_chunk<Mdl>();
    constexpr int NC = 1 + (Mdl::PATH_NCONST > Mdl::FIN_NCONST ? Mdl::PATH_NCONST : Mdl::FIN_NCONST), STRIDE = Mdl::PATH_NVAR | 1;
    constexpr int NCODE = NX * NX + NX * NU + NX * NP + NX * NX + NX * NU + NX * NP + NU * NU + NU * NP;   // entries of the 8 path families
    __shared__ double blk[NC + CH * STRIDE + Mdl::FIN_NVAR + 8];
    __shared__ short codes[NCODE > NX * NX + NX * NP ? NCODE : NX * NX + NX * NP];
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
    const int b = blockIdx.x / (nchunk + 1), c = blockIdx.x % (nchunk + 1), lane = threadIdx.x;
    double th[NP > 0 ? NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    if (lane == 0) blk[0] = 0.0;
    if (c == nchunk) {                                       // terminal matrices hxx, hxe at x_T (PDP.py:300-301)
        for (int i_ = lane; i_ < Mdl::FIN_NCONST; i_ += 64) blk[1 + i_] = Mdl::fin_const(i_);
        for (int i = lane; i < NX * NX; i += 64) codes[i] = (short)Mdl::fin_code(0, i);
        for (int i = lane; i < NX * NP; i += 64) codes[NX * NX + i] = (short)Mdl::fin_code(1, i);
        if (lane == 0) {
            double xT[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xT[i] = x[((int64_t)b * (T + 1) + T) * NX + i];
            PackedSink s{blk + NC};
            Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
        }
        __syncthreads();
        for (int m2 = 0; m2 < 2; ++m2) {
            double* dst = m2 == 0 ? o.hxx : o.hxe;
            const int rc = m2 == 0 ? NX * NX : NX * NP, co = m2 == 0 ? 0 : NX * NX;
            if (!dst) continue;
            for (int q = lane; q < rc; q += 64) {
                const int code = codes[co + q];
                dst[(int64_t)b * rc + q] = code >= 0 ? blk[NC + code] : blk[code == -1 ? 0 : 1 + (-2 - code)];
            }
        }
        return;
    }
    const int t0 = c * ch, cnt = min(ch, T - t0);
    for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
    {
        int base = 0;
#pragma unroll
        for (int mat = 0; mat < 8; ++mat) {
            const int rc = Mdl::PATH_ROWS[mat] * Mdl::PATH_COLS[mat];
            for (int i = lane; i < rc; i += 64) codes[base + i] = (short)Mdl::path_code(mat, i);
            base += rc;
        }
    }
    const int64_t bt0 = (int64_t)b * T + t0;
    if (lane < cnt) {
        const int64_t bt = bt0 + lane;
        double xc[NX], uc[NU], lc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = x[((int64_t)b * (T + 1) + t0 + lane) * NX + i]; lc[i] = lam[bt * NX + i]; }
#pragma unroll
        for (int i = 0; i < NU; ++i) uc[i] = u[bt * NU + i];
        PackedSink s{blk + NC + lane * STRIDE};
        Mdl::eval_path(xc, uc, lc, th, pc, s);
        if (o.dHu) {
            double hu[NU];
            Mdl::dHu(xc, uc, lc, th, pc, hu);
#pragma unroll
            for (int i = 0; i < NU; ++i) o.dHu[bt * NU + i] = hu[i];
        }
    }
    __syncthreads();
    constexpr int oF = 0, oG = oF + NX * NX, oE = oG + NX * NU, oHxx = oE + NX * NP, oHxu = oHxx + NX * NX, oHxe = oHxu + NX * NU,
                  oHuu = oHxe + NX * NP, oHue = oHuu + NU * NU;
    auxsys_expand<Mdl, 0>(blk, codes + oF, NC, STRIDE, cnt, o.dynF ? o.dynF + bt0 * NX * NX : nullptr, lane);
    auxsys_expand<Mdl, 1>(blk, codes + oG, NC, STRIDE, cnt, o.dynG ? o.dynG + bt0 * NX * NU : nullptr, lane);
    auxsys_expand<Mdl, 2>(blk, codes + oE, NC, STRIDE, cnt, o.dynE ? o.dynE + bt0 * NX * NP : nullptr, lane);
    auxsys_expand<Mdl, 3>(blk, codes + oHxx, NC, STRIDE, cnt, o.Hxx ? o.Hxx + bt0 * NX * NX : nullptr, lane);
    auxsys_expand<Mdl, 4>(blk, codes + oHxu, NC, STRIDE, cnt, o.Hxu ? o.Hxu + bt0 * NX * NU : nullptr, lane);
    auxsys_expand<Mdl, 5>(blk, codes + oHxe, NC, STRIDE, cnt, o.Hxe ? o.Hxe + bt0 * NX * NP : nullptr, lane);
    auxsys_expand<Mdl, 6>(blk, codes + oHuu, NC, STRIDE, cnt, o.Huu ? o.Huu + bt0 * NU * NU : nullptr, lane, o.Huu_damp ? o.Huu_damp[b] : 0.0);
    auxsys_expand<Mdl, 7>(blk, codes + oHue, NC, STRIDE, cnt, o.Hue ? o.Hue + bt0 * NU * NP : nullptr, lane);
    if (o.Hux) {                                             // Hux = Hxu' (the reference stores both, PDP.py:295)
        double* dst = o.Hux + bt0 * NU * NX;
        for (int q = lane; q < cnt * NU * NX; q += 64) {
            const int tl = q / (NU * NX), r = q - tl * (NU * NX), j = r / NX, i = r - j * NX;      // Hux[j][i] = Hxu[i][j]
            const int code = codes[oHxu + i * NU + j];
            dst[q] = code >= 0 ? blk[NC + tl * STRIDE + code] : blk[code == -1 ? 0 : 1 + (-2 - code)];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// OC: batched Newton solve (pdp_oc_solve_batched): per-sample bookkeeping between the costate / aux / LQR / rollout kernels.
// The iteration logic is the one of pdp_amd/ocsolver.py (which documents the choices); here it runs without the host.
// ------------------------------------------------------------------------------------------------------
struct OcSolveState {            // per-sample solver state in the workspace
    double *J, *mu, *gnorm;      // cost of the current trajectory, Levenberg-Marquardt damping, |H_u|_inf
    int32_t *newton, *converged, *lqr_status;
    int32_t* counters;           // [2]: number of converged samples of even / odd iterations (polled by the host)
};
PDP_DEV double wave_max_nan(double v) {                      // max over the wave; NaN if any lane holds NaN (like torch.amax)
    bool nan = v != v;
    v = nan ? 0.0 : v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return __any(nan) ? __builtin_nan("") : v;
}

// after the costates: stationarity residual, convergence, Newton / Gauss-Newton mode, costates used for the Hessians
template <class Mdl>
__global__ void __launch_bounds__(64) oc_newton_prepare_kernel(int B, int T, int it, double tol, double newton_switch, const double* __restrict__ u,
                                                                const double* __restrict__ dHu, const double* __restrict__ lam,
                                                                double* __restrict__ lam_eff, OcSolveState s) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x, lane = threadIdx.x;
    double g = 0.0, um = 0.0;
    for (int q = lane; q < T * NU; q += 64) {
        const double h = fabs(dHu[(int64_t)b * T * NU + q]), a = fabs(u[(int64_t)b * T * NU + q]);
        g = (h != h || g != g) ? __builtin_nan("") : fmax(g, h);
        um = (a != a || um != um) ? __builtin_nan("") : fmax(um, a);
    }
    g = wave_max_nan(g);
    const double scale = 1.0 + wave_max_nan(um);
    const bool conv = g <= tol * scale;
    // Newton once the residual is small relative to the controls (Gauss-Newton, always a descent direction, before), with
    // hysteresis: a sample whose residual has grown back by two orders of magnitude is no longer in Newton's basin
    const bool newton = (s.newton[b] != 0 && g <= 100.0 * newton_switch * scale) || g <= newton_switch * scale;
    for (int q = lane; q < T * NX; q += 64) lam_eff[(int64_t)b * T * NX + q] = newton ? lam[(int64_t)b * T * NX + q] : 0.0;   // GN: Hessians at lambda = 0
    if (lane == 0) {
        s.gnorm[b] = g; s.converged[b] = conv; s.newton[b] = newton;
        if (conv) atomicAdd(&s.counters[it & 1], 1);
        if (b == 0) s.counters[(it + 1) & 1] = 0;
    }
}

// closed-loop line search: one lane per (sample, trial); trial k uses alpha = 2^-k
template <class Mdl>
__global__ void __launch_bounds__(64) oc_linesearch_kernel(int B, int T, int K, const double* __restrict__ x0, const double* __restrict__ ubar, const double* __restrict__ xbar,
                                     const double* __restrict__ gains, const double* __restrict__ theta, int tb, double* __restrict__ xt,
                                     double* __restrict__ ut, double* __restrict__ Jt) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, GSZ = NX * NU + NU;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * K) return;
    const int b = idx / K, k = idx - b * K;
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    Jt[idx] = closed_loop_rollout<Mdl>(T, ldexp(1.0, -k), x0 + (int64_t)b * NX, ubar + (int64_t)b * T * NU, xbar + (int64_t)b * (T + 1) * NX,
                                       gains + (int64_t)b * T * GSZ, th, pc, xt + (int64_t)idx * (T + 1) * NX, ut + (int64_t)idx * T * NU);
}

// Armijo selection among the trials, acceptance of the step, damping / mode update (one wavefront per sample)
template <class Mdl>
__global__ void __launch_bounds__(64) oc_ls_select_kernel(int B, int T, int K, const double* __restrict__ dHu, const double* __restrict__ dU,
                                                           const double* __restrict__ xt, const double* __restrict__ ut, const double* __restrict__ Jt,
                                                           double* __restrict__ x, double* __restrict__ u, OcSolveState s) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x, lane = threadIdx.x;
    // first-order change of the cost along the open-loop direction, sum_t H_u' du, must be negative
    double sl = 0.0;
    bool fin = true;
    for (int q = lane; q < T * NU; q += 64) {
        const double d = dU[(int64_t)b * T * NU + q];
        sl += dHu[(int64_t)b * T * NU + q] * d;
        fin = fin && fabs(d) <= 1.7e308;
    }
    sl = wave_sum(sl);
    const bool conv = s.converged[b] != 0, newton = s.newton[b] != 0;
    const bool bad = s.lqr_status[b] != 0 || !__all(fin) || !(sl < 0.0);
    const double J = s.J[b];
    int pick = -1;
    if (!conv && !bad) {
        for (int k = 0; k < K; ++k) {
            const double a = ldexp(1.0, -k), Jk = Jt[(int64_t)b * K + k];
            if (fabs(Jk) <= 1.7e308 && Jk <= J + 1e-4 * a * sl + 1e-13 * fabs(J)) { pick = k; break; }      // Armijo, up to rounding of J
        }
    }
    if (pick >= 0) {
        const double* xs = xt + ((int64_t)b * K + pick) * (T + 1) * NX;
        const double* us = ut + ((int64_t)b * K + pick) * T * NU;
        for (int q = lane; q < (T + 1) * NX; q += 64) x[(int64_t)b * (T + 1) * NX + q] = xs[q];
        for (int q = lane; q < T * NU; q += 64) u[(int64_t)b * T * NU + q] = us[q];
    }
    if (lane == 0) {
        const bool failed = pick < 0 && !conv;                // no acceptable step (bad direction included)
        if (pick >= 0) s.J[b] = Jt[(int64_t)b * K + pick];
        // failure: Newton falls back to Gauss-Newton, Gauss-Newton raises its damping; success: relax the damping
        double mu = s.mu[b];
        if (failed && !newton) mu = fmax(mu * 4.0, 1e-4);
        else if (!failed) mu *= 0.5;
        if (mu < 1e-8) mu = 0.0;
        s.mu[b] = mu;
        s.newton[b] = newton && !failed;
    }
}

// ------------------------------------------------------------------------------------------------------
// OC: fused forward + costates + aux system (LDS) + Riccati + PDP gradient, one wavefront per trajectory
// ------------------------------------------------------------------------------------------------------
typedef unsigned pdp_u2x __attribute__((ext_vector_type(2)));
typedef unsigned pdp_u4 __attribute__((ext_vector_type(4)));
struct Gather { int off[4]; int tmul[4]; };
// running form: cur[r] is the ABSOLUTE LDS byte address of the element for the current step (the base of the dynamic LDS block is a
// link-time constant the compiler cannot fold: added once here, not as a VALU add in front of every ds_read); the ds_read / ds_write
// address is the register itself
#define PDP_LDS __attribute__((address_space(3)))
PDP_DEV unsigned lds_addr(const double* p) { return (unsigned)(uintptr_t)(PDP_LDS const double*)p; }
struct GatherRun { unsigned cur[4]; int tmul[4]; };
PDP_DEV GatherRun gather_at(const Gather& g, int tl, const double* lds) {
    GatherRun r;
    const unsigned base = lds_addr(lds);
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.cur[k] = base + 8u * (unsigned)(g.off[k] + tl * g.tmul[k]); r.tmul[k] = 8 * g.tmul[k]; }
    return r;
}
PDP_DEV void scatter_run(GatherRun& g, const d4 v, int dir) {      // write, then advance
#pragma unroll
    for (int k = 0; k < 4; ++k) { *(PDP_LDS double*)(uintptr_t)g.cur[k] = v[k]; g.cur[k] += (unsigned)(dir * g.tmul[k]); }
}
template <int NR = 4>
PDP_DEV d4 gather_run(GatherRun& g, int dir) {      // read, then advance by dir (+1 / -1) time steps
    d4 v = zero4();
#pragma unroll
    for (int k = 0; k < NR; ++k) { v[k] = *(PDP_LDS const double*)(uintptr_t)g.cur[k]; g.cur[k] += (unsigned)(dir * g.tmul[k]); }
    return v;
}

// offsets (in doubles, relative to the LDS block [cpool | pool]) of tile element (lane, r)
template <class CodeFn>
PDP_DEV void make_gather(Gather& g, int lane, int cpool_n, int stride, CodeFn code_of /* (row, col) -> code or -1 */) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int code = code_of(tile_row(lane, r), tile_col(lane));
        if (code >= 0) { g.off[r] = cpool_n + code; g.tmul[r] = stride; }
        else { g.off[r] = (code == -1) ? 0 : (1 + (-2 - code)); g.tmul[r] = 0; }
    }
}
PDP_DEV d4 gather_tile(const double* lds, const Gather& g, int tl) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = lds[g.off[r] + tl * g.tmul[r]];
    return v;
}

// feedback gains of one step in the workspace: K [NU x NX], k [NU x NP], and one slot that receives (and hands back) the zeros of
// the tile elements outside those blocks
template <class Mdl>
__host__ __device__ constexpr int fused_gain0_doubles() { return Mdl::NX * Mdl::NU + Mdl::NU * Mdl::NP + 1; }      // K | k | zero sink
// Riccati record of a stage (optional output of the fused gradient unit): P_{t+1} [NX x NX] | W_{t+1} [NX x NP] | one scratch word (zero sink of the tile stores)
template <class Mdl>
__host__ __device__ constexpr int oc_riccati_doubles() { return Mdl::NX * Mdl::NX + Mdl::NX * Mdl::NP + 1; }
// Prediction record of a stage (optional fp32 output of the fused gradient unit): everything the first-order prediction of the next OC solve's starting point
// needs, packed and in SINGLE precision - X_{t+1} [NX x NP] | U_t [NU x NP] | upper triangle of P_{t+1} (row-major packed) | W_{t+1} [NX x NP].  A starting
// point is only ever first-order accurate in dtheta; fp32 factors (6e-8 relative on a correction of size |dtheta|) are far below that error, and the record
// is 74 MB per launch at C3 where the fp64 outputs (dxdp, dudp, riccati) are 181 MB - the difference between a prediction that is read at HBM speed inside the
// solver (42 us of a 265 us solve) and one that stays in the Infinity Cache (profiles/r04_predict_cost.txt).
template <class Mdl>
struct PredRec {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    static constexpr int X = 0, U = NX * NP, P = U + NU * NP, W = P + NX * (NX + 1) / 2, SIZE = W + NX * NP;
    __host__ __device__ static constexpr int tri(int i, int j) { return i <= j ? i * NX - i * (i - 1) / 2 + (j - i) : j * NX - j * (j - 1) / 2 + (i - j); }
};
struct PredMap { unsigned voff[4]; };        // BYTE offsets of a tile's elements inside one stage's record (fp32), or out of range
template <class Fn>
PDP_DEV PredMap pred_map(int lane, Fn idx /* (row, col) -> float index inside the record, or -1 */) {
    PredMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int k = idx(tile_row(lane, r), tile_col(lane)); m.voff[r] = k >= 0 ? 4u * (unsigned)k : 0x80000000u; }
    return m;
}
template <int NR = 4, class RS>
PDP_DEV void pred_store(RS rs, unsigned soff, const PredMap& m, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float x = (float)v[r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, m.voff[r], soff, 0);
    }
}
template <class Mdl>
struct PredMaps {
    PredMap X, U, P, W;
    PDP_DEV explicit PredMaps(int lane) {
        using R = PredRec<Mdl>;
        constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
        X = pred_map(lane, [](int r, int c) { return (r < NX && c >= NU && c < NU + NP) ? R::X + r * NP + (c - NU) : -1; });
        U = pred_map(lane, [](int r, int c) { return (r < NU && c >= NU && c < NU + NP) ? R::U + r * NP + (c - NU) : -1; });
        P = pred_map(lane, [](int r, int c) { return (r <= c && c < NX) ? R::P + R::tri(r, c) : -1; });
        W = pred_map(lane, [](int r, int c) { return (r < NX && c >= NU && c < NU + NP) ? R::W + r * NP + (c - NU) : -1; });
    }
};
template <class Mdl>
__host__ __device__ constexpr int fused_gain_doubles() {
    return fused_gain0_doubles<Mdl>();
}

// experiment hooks (probes/occupancy_variant.py): -DPDP_FUSED_CHUNK=<steps per chunk> shrinks the LDS pool, -DPDP_FUSED_WAVES=<n> asks the
// register allocator for n waves per SIMD (n = 2: 256 registers per wave, VGPRs + AGPRs together)
#ifdef PDP_FUSED_CHUNK
template <class Mdl> struct FusedChunk { static constexpr int value = PDP_FUSED_CHUNK; };
#else
template <class Mdl> struct FusedChunk { static constexpr int value = Mdl::CHUNK; };
#endif
#ifdef PDP_FUSED_WAVES
#define PDP_FUSED_OCCUPANCY __attribute__((amdgpu_waves_per_eu(PDP_FUSED_WAVES, PDP_FUSED_WAVES)))
#else
#define PDP_FUSED_OCCUPANCY
#endif

template <class Mdl>
struct FusedLayout {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = FusedChunk<Mdl>::value;
    static constexpr int NCB = Mdl::PATHA_NCONST + Mdl::PATHB_NCONST;          // constants of the two backward groups, A first
    static constexpr int NC = 1 + (NCB > Mdl::FWD_NCONST ? (NCB > Mdl::FIN_NCONST ? NCB : Mdl::FIN_NCONST)
                                                         : (Mdl::FWD_NCONST > Mdl::FIN_NCONST ? Mdl::FWD_NCONST : Mdl::FIN_NCONST));
    static constexpr int NA = Mdl::PATHA_NVAR, NB = Mdl::PATHB_NVAR;           // backward pool row: [patha | pathb | lambda_{t+1} (NX)]
    static constexpr int LAM = NA + NB;
    static constexpr int BSTRIDE = (NA + NB + NX) | 1;                         // odd -> conflict-free
    static constexpr int FEXTRA = NX + NU;                                      // x - x_demo, u - u_demo per step
    static constexpr int FSTRIDE = (Mdl::FWD_NVAR + FEXTRA) | 1;
    static constexpr int POOL = CH * (BSTRIDE > FSTRIDE ? BSTRIDE : FSTRIDE) > Mdl::FIN_NVAR + 1 ? CH * (BSTRIDE > FSTRIDE ? BSTRIDE : FSTRIDE) : Mdl::FIN_NVAR + 1;
};

// pool size in doubles: the aux-matrix pool, or the x/u staging of the rollout phase, whichever is larger
template <class Mdl>
__host__ __device__ inline int fused_pool_doubles(int T) {
    const int stage = (T + 1) * Mdl::NX + T * Mdl::NU;
    return FusedLayout<Mdl>::POOL > stage ? FusedLayout<Mdl>::POOL : stage;
}
template <class Mdl>
__host__ __device__ inline size_t fused_lds_bytes(int T) {
    return sizeof(double) * (size_t)(RICCATI_SCRATCH + FusedLayout<Mdl>::NC + fused_pool_doubles<Mdl>(T) + Mdl::NX + Mdl::NP + Mdl::NPC + 8);
}

template <class Mdl, bool RIC = false>
__global__ void __launch_bounds__(64) PDP_FUSED_OCCUPANCY oc_pdp_fused_kernel(int B, int T, int flags, const double* __restrict__ x0, const double* __restrict__ u,
                                                           const double* __restrict__ theta, int tb, const double* __restrict__ demo_x,
                                                           const double* __restrict__ demo_u, double* __restrict__ x, double* __restrict__ lam,
                                                           double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ dxdp,
                                                           double* __restrict__ dudp, int32_t* __restrict__ status, double* __restrict__ ws_gain,
                                                           double* __restrict__ riccati, float* __restrict__ prec) {
    using L = FusedLayout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = L::CH, M = NU;
    constexpr int GSZ = fused_gain_doubles<Mdl>();         // per step: K [NU x NX] | k [NU x NP] | zero sink
    // SMALL (n <= 4: pendulum, cart-pole, robot arm): every matrix of the recursion fits the rows-0..3 register of its tile and every
    // product is ONE v_mfma_f64_4x4x4_4b (pdp_riccati_small.h) instead of four 64-cycle 16x16x4 MFMAs on a tile that is 94 % padding:
    // left operands are gathered in "rep" form (the 4 x 4 block replicated in the four column blocks), only register 0 of a tile is live
    constexpr bool SMALL = NX <= 4;
    constexpr int NRT = SMALL ? 1 : 4;                  // live registers of an n-row tile
    constexpr int GSZ0 = fused_gain0_doubles<Mdl>();
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* scratch = lds;                              // RICCATI_SCRATCH
    double* blk = lds + RICCATI_SCRATCH;                // [cpool (NC) | pool]
    double* pool = blk + L::NC;
    double* dlT = pool + fused_pool_doubles<Mdl>(T);    // x_T - xdemo_T (NX)
    const int b = blockIdx.x, lane = threadIdx.x;
    const int tlane = small_transpose_lane(lane);
    const d4 z = zero4();
    // theta and the theta-only precomputed values are parked in LDS and re-read inside every block of generated scalar code:
    // kept in registers they would occupy 2 (NP + NPC) VGPRs for the whole kernel, which sits at the 256-VGPR ceiling
    double* par = dlT + NX;                             // [theta (NP) | pc (NPC)]
    {
        double th0[NP], pc0[Mdl::NPC];
        load_theta<Mdl>(theta, b, tb, th0);
        Mdl::precompute(th0, pc0);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NP; ++i) par[i] = th0[i];
#pragma unroll
            for (int i = 0; i < Mdl::NPC; ++i) par[NP + i] = pc0[i];
        }
        wave_lds_sync();
    }
#define PDP_LOAD_PAR()                                              \
    double th[NP], pc[Mdl::NPC];                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) th[i_] = par[i_]; \
    _Pragma("unroll") for (int i_ = 0; i_ < Mdl::NPC; ++i_) pc[i_] = par[NP + i_]
    double* xb = x + (int64_t)b * (T + 1) * NX;
    double* lb = lam + (int64_t)b * T * NX;
    const double* ub = u + (int64_t)b * T * NU;
    double* gw = ws_gain + (int64_t)b * T * GSZ;
#ifdef PDP_PHASE_TIMING
    long long tstamp[8]; int nst = 0;
#define PDP_STAMP() tstamp[nst++] = __builtin_readcyclecounter()
    long long fine[16]; for (int i = 0; i < 16; ++i) fine[i] = 0;
#define PDP_FINE(i, cond) if (cond) fine[i] = __builtin_readcyclecounter()
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;           // per-phase cycle totals over all chunks
#define PDP_ACC0() tlast = __builtin_readcyclecounter()
#define PDP_ACC(k) { long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - tlast; tlast = now_; }
#else
#define PDP_STAMP()
#define PDP_FINE(i, cond)
#define PDP_ACC0()
#define PDP_ACC(k)
#endif
    PDP_STAMP();

    // ---------------- phase R: trajectory x+ = f(x,u,theta) (scalar recursion, executed uniformly by all lanes; u staged in the
    // still unused LDS pool).  The costates are NOT computed here: they are propagated on MFMA
    // inside the backward chunks (lambda_t = c_x + F_t' lambda_{t+1} with the F_t tiles the Riccati step gathers anyway).
    if (!(flags & PDP_OC_GIVEN_TRAJ)) {
        double* us = pool;                                   // T x NU
        for (int i = lane; i < T * NU; i += 64) us[i] = ub[i];
        PDP_LOAD_PAR();
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xb[i] = xc[i];
        }
        wave_lds_sync();
        const auto rsX = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((T + 1) * NX * 8), 0x00020000);
        const unsigned xvoff = lane == 0 ? 0u : 0x80000000u;
        PDP_ACC0();
        double un[NU];                                       // u_{t+1} is read from LDS while step t computes
#pragma unroll
        for (int i = 0; i < NU; ++i) un[i] = us[i];
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
            Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            // x_{t+1} goes straight to the API output from lane 0: global stores are counted by vmcnt, which nothing in the loop waits
            // for; LDS stores would sit in front of the next step's u reads in the in-order LDS counter (probes/rollout_probe.hip).  No
            // branch either: a buffer store whose offset is out of range in every other lane (a conditional block makes the waits that
            // follow it conservative).
            {
                const unsigned so = (unsigned)((t + 1) * NX) * 8u;
#pragma unroll
                for (int i = 0; i + 1 < NX; i += 2) {
                    pdp_u4 w;
                    w.x = (unsigned)__double2loint(xn[i]); w.y = (unsigned)__double2hiint(xn[i]);
                    w.z = (unsigned)__double2loint(xn[i + 1]); w.w = (unsigned)__double2hiint(xn[i + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsX, xvoff + 8u * i, so, 0);
                }
                if constexpr (NX & 1) {
                    pdp_u2x w;
                    w.x = (unsigned)__double2loint(xn[NX - 1]); w.y = (unsigned)__double2hiint(xn[NX - 1]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsX, xvoff + 8u * (NX - 1), so, 0);
                }
            }
        }
        PDP_ACC(6);
        __threadfence_block();                               // x is re-read below by other lanes of this wave
        wave_lds_sync();
    }
    PDP_STAMP();

    // ---------------- terminal condition: P = hxx(x_T), W = hxe(x_T) ----------------------------------
    bool ok = true;
    d4 P, W2;
    {
        if (lane == 0) blk[0] = 0.0;
        for (int i_ = lane; i_ < Mdl::FIN_NCONST; i_ += 64) blk[1 + i_] = Mdl::fin_const(i_);
        if (lane == 0) {
            PDP_LOAD_PAR();
            double xT[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xT[i] = xb[T * NX + i];
            PackedSink s{pool};
            Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
        }
        wave_lds_sync();
        Gather gP, gW;
        make_gather(gP, lane, L::NC, 0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::fin_code(0, r * NX + (c & 3)) : -1)
                                                                         : ((r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1); });
        make_gather(gW, lane, L::NC, 0, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fin_code(1, r * NP + (c - M)) : -1; });
        P = gather_tile(blk, gP, 0);
        W2 = gather_tile(blk, gW, 0);
        wave_lds_sync();
    }
    PDP_STAMP();

    // ---------------- backward sweep: chunks of CH time steps ------------------------------------------
    // per chunk: (A) lane = time step evaluates F, G, E, c_x  ->  (C) costates through the chunk on MFMA: lambda_t = c_x + F_t' lambda_{t+1}
    //            (B) lane = time step evaluates the lambda-weighted Hessians Hxx, Hxu, Hxe, Huu, Hue  ->  Riccati steps
    {
        for (int i_ = lane; i_ < Mdl::PATHA_NCONST; i_ += 64) blk[1 + i_] = Mdl::patha_const(i_);
        for (int i_ = lane; i_ < Mdl::PATHB_NCONST; i_ += 64) blk[1 + Mdl::PATHA_NCONST + i_] = Mdl::pathb_const(i_);
        constexpr int NA = L::NA, NCA = Mdl::PATHA_NCONST;
        auto codeA = [](int mat, int i) { return Mdl::patha_code(mat, i); };
        auto codeB = [](int mat, int i) { int c = Mdl::pathb_code(mat, i); return c >= 0 ? c + NA : (c == -1 ? -1 : c - NCA); };
        Gather gF, gY, gHxx, gHX, gHU, gCX;
        make_gather(gF, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeA(0, r * NX + (c & 3)) : -1)
                                                                                  : ((r < NX && c < NX) ? codeA(0, r * NX + c) : -1); });
        make_gather(gY, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
            return r >= NX ? -1 : (c < M ? codeA(1, r * NU + c) : (c < M + NP ? codeA(2, r * NP + (c - M)) : -1)); });
        make_gather(gCX, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && c == 0) ? codeA(3, r) : -1; });
        Gather gGr, gHux;                                 // G replicated in the four column blocks (operand of the 4-row products); Hux = Hxu'
        make_gather(gGr, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeA(1, r * NU + (c & 3)) : -1; });
        make_gather(gHux, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? codeB(1, (c & 3) * NU + r) : -1)
                                                                                    : ((r < M && c < NX) ? codeB(1, c * NU + r) : -1); });
        make_gather(gHxx, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeB(0, r * NX + (c & 3)) : -1)
                                                                                    : ((r < NX && c < NX) ? codeB(0, r * NX + c) : -1); });
        make_gather(gHX, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
            return r >= NX ? -1 : (c < M ? codeB(1, r * NU + c) : (c < M + NP ? codeB(2, r * NP + (c - M)) : -1)); });
        make_gather(gHU, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
            return r >= M ? -1 : (c < M ? codeB(3, r * NU + c) : (c < M + NP ? codeB(4, r * NP + (c - M)) : -1)); });
        // gains of a step in the workspace: K [NU x NX] (rows 0..3 of its tile: one register), k [NU x NP], zero sink
        const TileMapBytes mK = make_tile_map_sink(NU, NX, NX, 0, 0, lane, GSZ0 - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
        // Riccati record of a stage (RIC; see oc_pdp_fused3_kernel): P_{t+1} [NX x NX] | W_{t+1} [NX x NP] | zero sink
        constexpr int RSZ = oc_riccati_doubles<Mdl>();
        [[maybe_unused]] const TileMapBytes mRP = make_tile_map_sink(NX, NX, NX, 0, 0, lane, RSZ - 1), mRW = make_tile_map_sink(NX, NP, NP, 0, M, lane, NX * NP);
        [[maybe_unused]] double* rw = (RIC && riccati) ? riccati + (int64_t)b * T * RSZ : nullptr;
        // prediction record (RIC, fp32, see PredRec): range-checked buffer stores, a NULL record is a resource of size 0
        [[maybe_unused]] const PredMaps<Mdl> pm(lane);
        [[maybe_unused]] const bool precPW = RIC && prec && !(flags & PDP_OC_RECORD_PRIMAL);      // (PDP_OC_RECORD_PRIMAL: the P | W stores go to a resource of size 0)
        [[maybe_unused]] const auto rsPR = __builtin_amdgcn_make_buffer_rsrc((void*)(precPW ? (void*)(prec + (int64_t)b * T * PredRec<Mdl>::SIZE) : (void*)ws_gain), 0,
                                                                              precPW ? (int)((int64_t)T * PredRec<Mdl>::SIZE * 4) : 0, 0x00020000);
        const bool given = (flags & PDP_OC_GIVEN_TRAJ) != 0;
        // costate tile: column 0 holds lambda_{t+1}; terminal value lambda_T = h_x(x_T)
        d4 Lam = z;
        if (!given) {
            PDP_LOAD_PAR();
            double xT[NX], lT[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xT[i] = xb[T * NX + i];
            Mdl::dhx(xT, th, pc, lT);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) dlT[i] = lT[i];
            }
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX && tile_col(lane) == 0) Lam[r] = dlT[row]; }
        }
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
        for (int c = nchunk - 1; c >= 0; --c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            PDP_ACC0();
            wave_lds_sync();
            if (lane < cnt) {                       // (A) lane = time step: F, G, E, c_x at (x_t, u_t)
                PDP_LOAD_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU];
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xb[t * NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                PackedSink s{pool + lane * L::BSTRIDE};
                Mdl::eval_patha(xc, uc, nullptr, th, pc, s);
            }
            wave_lds_sync();
            PDP_ACC(0);
            if (!given) {                           // (C) costates through the chunk; pool row tl receives lambda_{t+1}
                // F_t and c_x,t do not depend on the recursion: they are gathered one step ahead of the MFMA chain that needs them
                // (not below the first row of the pool: the last step of a chunk issues no prefetch).
                GatherRun cF = gather_at(gF, cnt - 1, blk), cC = gather_at(gCX, cnt - 1, blk), wL;
#pragma unroll
                for (int r = 0; r < 4; ++r) {       // lambda_{t+1} goes to pool row tl; tile elements outside column 0 to a dead slot
                    const int row = tile_row(lane, r);
                    const bool valid = tile_col(lane) == 0 && row < NX;
                    wL.cur[r] = lds_addr(lds) + (valid ? 8u * (unsigned)((pool - lds) + (cnt - 1) * L::BSTRIDE + L::LAM + row) : 0u);     // scratch[0]: idle here
                    wL.tmul[r] = valid ? 8 * L::BSTRIDE : 0;
                }
                d4 Fc = gather_run<NRT>(cF, -1), CX = gather_run<NRT>(cC, -1);
                auto cstep = [&](int tl, const d4 Lin, d4& Lout) {
                    d4 Fc_n = Fc, CX_n = CX;
                    if (tl > 0) { Fc_n = gather_run<NRT>(cF, -1); CX_n = gather_run<NRT>(cC, -1); }
                    scatter_run(wL, Lin, -1);
                    if constexpr (SMALL) { Lout = z; Lout[0] = mma4_blk(Fc[0], Lin[0], CX[0]); }
                    else Lout = mma_tn(Fc, Lin, CX);     // lambda_t = c_x(x_t,u_t) + F_t' lambda_{t+1}
                    Fc = Fc_n; CX = CX_n;
                };
                // two steps per trip with the costate tile alternating between two register sets: the MFMA chain of a step reads
                // its predecessor's tile until its last instruction, so one set would need a copy in the middle of every chain
                d4 Lam2 = z;
                int tl = cnt - 1;
                for (; tl >= 1; tl -= 2) { cstep(tl, Lam, Lam2); cstep(tl - 1, Lam2, Lam); }
                if (tl == 0) { cstep(0, Lam, Lam2); Lam = Lam2; }
                wave_lds_sync();
            }
            PDP_ACC(1);
            if (lane < cnt) {                       // (B) lane = time step: Hamiltonian Hessians at (x_t, u_t, lambda_{t+1})
                PDP_LOAD_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU], lc[NX];
                double* row = pool + lane * L::BSTRIDE;
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xb[t * NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                if (given) {
#pragma unroll
                    for (int i = 0; i < NX; ++i) lc[i] = lb[t * NX + i];
                } else {
#pragma unroll
                    for (int i = 0; i < NX; ++i) { lc[i] = row[L::LAM + i]; lb[t * NX + i] = lc[i]; }      // costate is an API output
                }
                PackedSink s{row + NA};
                Mdl::eval_pathb(xc, uc, lc, th, pc, s);
            }
            wave_lds_sync();
            PDP_ACC(2);
            // operands of step tl are gathered one step ahead (the pool is read-only inside the chunk); running LDS offsets
            GatherRun rF = gather_at(gF, cnt - 1, blk), rY = gather_at(gY, cnt - 1, blk), rHxx = gather_at(gHxx, cnt - 1, blk), rHX = gather_at(gHX, cnt - 1, blk),
                      rHU = gather_at(gHU, cnt - 1, blk), rGr = gather_at(gGr, cnt - 1, blk), rHux = gather_at(gHux, cnt - 1, blk);
            // F and [G|E] feed the first MFMAs of a step and are gathered one step ahead; the Hessian tiles are accumulator inputs
            // of later MFMAs: their reads are issued at the top of the step, straight into the accumulator registers.  The last step
            // of a chunk prefetches nothing (no LDS read outside the pool).  Two steps per trip with the prefetched tiles alternating
            // between two register sets (no copies at the back edge).
#ifndef PDP_FUSED_NO_PREFETCH
            d4 Fa = gather_run<NRT>(rF, -1), Ya = gather_run<NRT>(rY, -1), Fb = z, Yb = z;
#else
            d4 Fa = z, Ya = z, Fb = z, Yb = z;
#endif
            auto bstep = [&](int tl, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {
                const int t = t0 + tl;
                PDP_FINE(0, t == 20);
                d4 Hxx = gather_run<NRT>(rHxx, -1), HX2 = gather_run<NRT>(rHX, -1), HU2 = gather_run<1>(rHU, -1), Grep = gather_run<NRT>(rGr, -1), Hux = gather_run<1>(rHux, -1);
#ifndef PDP_FUSED_NO_PREFETCH
                if (tl > 0) { Fn = gather_run<NRT>(rF, -1); Yn = gather_run<NRT>(rY, -1); }
                const d4 Fu = Fc, Yu = Yc;
#else       // two waves per SIMD hide the gather latency behind each other: no one-step-ahead copies of F and [G|E] (16 registers)
                const d4 Fu = gather_run<NRT>(rF, -1), Yu = gather_run<NRT>(rY, -1);
                (void)Fc; (void)Yc; (void)Fn; (void)Yn;
#endif
                PDP_FINE(1, t == 20);
                if constexpr (SMALL) {
                    SmallGains g;
                    double Pr = P[0], Wr = W2[0];
                    if constexpr (RIC) {                     // (P is in rep form: the first column block is P itself, the replicas must not reach the sink slot)
                        d4 Pt_ = z, Wt_ = z;
                        Pt_[0] = (lane & 12) == 0 ? Pr : 0.0; Wt_[0] = Wr;
                        if (rw) { store_all<1>(rw + t * RSZ, mRP, Pt_); store_all<1>(rw + t * RSZ + NX * NX, mRW, Wt_); }
                        pred_store<1>(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.P, Pt_);
                        pred_store<1>(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.W, Wt_);
                    }
                    ok = riccati_small_backward<M, true>(Pr, Wr, Fu[0], Yu[0], Grep[0], Hxx[0], HX2[0], HU2[0], Hux[0], lane, tlane, NP, g) && ok;
                    P[0] = Pr; W2[0] = Wr;
                    d4 Kt = z, IKt = z;
                    Kt[0] = (lane & 12) == 0 ? g.K : 0.0;      // K arrives in rep form: its first column block is K [NU x NX]; the replicas must not
                    IKt[0] = g.IK;                              // reach the zero sink slot the forward sweep reads back for absent elements
                    store_all<1>(gw + t * GSZ, mK, Kt);
                    store_all<1>(gw + t * GSZ + NX * NU, mIK, IKt);
                } else {
                RiccatiGains g;
                d4 P_old;
                if constexpr (RIC) {
                    if (rw) { store_all(rw + t * RSZ, mRP, P); store_all(rw + t * RSZ + NX * NX, mRW, W2); }
                    pred_store(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.P, P);
                    pred_store(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.W, W2);
                }
                ok = riccati_backward<M, false>(P, W2, Fu, Yu, Grep, Hxx, HX2, HU2, Hux[0], scratch, lane, NP, g, P_old) && ok;
                PDP_FINE(2, t == 20);
                store_all<1>(gw + t * GSZ, mK, g.K);
                store_all<1>(gw + t * GSZ + NX * NU, mIK, g.IK);
                }
                PDP_FINE(3, t == 20);
                PDP_FINE(4, t == 19);
            };
            int tl = cnt - 1;
            for (; tl >= 1; tl -= 2) { bstep(tl, Fa, Ya, Fb, Yb); bstep(tl - 1, Fb, Yb, Fa, Ya); }
            if (tl == 0) bstep(0, Fa, Ya, Fb, Yb);
            PDP_ACC(3);
        }
    }
    bool finite = tile_finite(P) && tile_finite(W2);
    PDP_STAMP();

    // ---------------- forward sweep: sensitivities X_t = dx_t/dtheta, U_t, loss and gradient -----------
    double acc = 0.0, lsum = 0.0;
    {
        wave_lds_sync();
        for (int i_ = lane; i_ < Mdl::FWD_NCONST; i_ += 64) blk[1 + i_] = Mdl::fwd_const(i_);
        constexpr int DLX = Mdl::FWD_NVAR, DLU = Mdl::FWD_NVAR + NX;      // pool slots of x - x_demo, u - u_demo
        Gather gFT, gGT, gE, gDX, gDU;
        make_gather(gFT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::fwd_code(0, (c & 3) * NX + r) : -1)
                                                                               : ((r < NX && c < NX) ? Mdl::fwd_code(0, c * NX + r) : -1); });
        make_gather(gGT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? Mdl::fwd_code(1, (c & 3) * NU + r) : -1)
                                                                               : ((r < M && c < NX) ? Mdl::fwd_code(1, c * NU + r) : -1); });
        make_gather(gE, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fwd_code(2, r * NP + (c - M)) : -1; });
        make_gather(gDX, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX) ? DLX + r : -1; });
        make_gather(gDU, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < M) ? DLU + r : -1; });
        const double* dxb = demo_x + (int64_t)b * (T + 1) * NX;
        const double* dub = demo_u + (int64_t)b * T * NU;
        d4 X2 = z;
        // feedback gains of step t are fetched one step ahead (each lane re-reads exactly what it stored)
        // K is read back transposed and replicated in the four column blocks (operand form of the 4-row product U = -K X - k)
        const TileMapBytes mKT = to_bytes_sink(make_rep4_map_transposed(NX, NU, NX, lane), GSZ0 - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
        d4 KTn = -load_all<NRT>(gw, mKT);
        d4 kn = -load_all<1>(gw + NX * NU, mIK);
        [[maybe_unused]] const PredMaps<Mdl> pmf(lane);
        [[maybe_unused]] const auto rsPRf = __builtin_amdgcn_make_buffer_rsrc((void*)(RIC && prec ? (void*)(prec + (int64_t)b * T * PredRec<Mdl>::SIZE) : (void*)ws_gain), 0,
                                                                               RIC && prec ? (int)((int64_t)T * PredRec<Mdl>::SIZE * 4) : 0, 0x00020000);
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            PDP_ACC0();
            wave_lds_sync();
            if (lane < cnt) {
                PDP_LOAD_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU];
                double* row = pool + lane * L::FSTRIDE;
#pragma unroll
                for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; double d = xc[i] - dxb[t * NX + i]; row[DLX + i] = d; lsum += d * d; }
#pragma unroll
                for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; double d = uc[i] - dub[t * NU + i]; row[DLU + i] = d; lsum += d * d; }
                PackedSink s{row};
                Mdl::eval_fwd(xc, uc, nullptr, th, pc, s);
            }
            wave_lds_sync();
            PDP_ACC(4);
            GatherRun rFT = gather_at(gFT, 0, blk), rGT = gather_at(gGT, 0, blk), rE = gather_at(gE, 0, blk), rDX = gather_at(gDX, 0, blk), rDU = gather_at(gDU, 0, blk);
            auto fstep = [&](int tl, const d4 Xc, d4& Xn, const d4 KTc, const d4 kc, d4& KTnx, d4& knx) {
                const int t = t0 + tl, tnx = (t + 1 < T) ? t + 1 : t;
                PDP_FINE(8, t == 20);
                KTnx = -load_all<NRT>(gw + tnx * GSZ, mKT);
                knx = -load_all<1>(gw + tnx * GSZ + NX * NU, mIK);
                d4 FT = gather_run<NRT>(rFT, 1);
                d4 GT = gather_run<1>(rGT, 1);
                d4 E2 = gather_run<NRT>(rE, 1);
                d4 DX = gather_run<NRT>(rDX, 1);            // (x_t - xd_t)[row] broadcast over columns
                d4 DU = gather_run<1>(rDU, 1);
                d4 U2;
                PDP_FINE(9, t == 20);
                if constexpr (SMALL) {
                    U2 = z; Xn = z;
                    double u0, x1;
                    riccati_small_forward(KTc[0], kc[0], FT[0], GT[0], E2[0], Xc[0], u0, x1);
                    U2[0] = u0; Xn[0] = x1;
                } else
                riccati_forward(KTc, kc, FT, GT, E2, Xc, U2, Xn);
                PDP_FINE(10, t == 20);
                acc += DX[0] * Xc[0] + DX[1] * Xc[1] + DX[2] * Xc[2] + DX[3] * Xc[3] + DU[0] * U2[0];
                if (dxdp) store_dense(dxdp + ((int64_t)b * (T + 1) + t) * NX * NP, NX, NP, NP, 0, M, lane, Xc);
                if (dudp) store_dense(dudp + ((int64_t)b * T + t) * NU * NP, NU, NP, NP, 0, M, lane, U2);
                if constexpr (RIC) {
                    pred_store<NRT>(rsPRf, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pmf.X, Xn);      // X_{t+1}
                    pred_store<1>(rsPRf, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pmf.U, U2);
                }
                PDP_FINE(11, t == 20);
                PDP_FINE(12, t == 21);
            };
            // two steps per trip, the sensitivity tile and the prefetched gains alternating between two register sets
            d4 Xb, KTb, kb;
            int tl = 0;
            for (; tl + 1 < cnt; tl += 2) { fstep(tl, X2, Xb, KTn, kn, KTb, kb); fstep(tl + 1, Xb, X2, KTb, kb, KTn, kn); }
            if (tl < cnt) { fstep(tl, X2, Xb, KTn, kn, KTb, kb); X2 = Xb; KTn = KTb; kn = kb; }
            PDP_ACC(5);
        }
        // terminal term (x_T - xd_T)' X_T   (cartpole_PDP.py:74)
        wave_lds_sync();
        if (lane < NX) { double d = xb[T * NX + lane] - dxb[T * NX + lane]; dlT[lane] = d; lsum += d * d; }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc += dlT[row] * X2[r]; }
        if (dxdp) store_dense(dxdp + ((int64_t)b * (T + 1) + T) * NX * NP, NX, NP, NP, 0, M, lane, X2);
        finite = finite && tile_finite(X2);
    }
    acc = sum_over_rowgroups(acc);
    lsum = wave_sum(lsum);
    // PDP_OC_PACKED: grad is [B][NP + 1] with the loss in the last column - the row the data-parallel iteration all-gathers
    const int gstride = (flags & PDP_OC_PACKED) ? NP + 1 : NP;
    if (lane >= M && lane < M + NP) grad[(int64_t)b * gstride + (lane - M)] = acc;
    if (lane == 0) { loss[b] = lsum; if (flags & PDP_OC_PACKED) grad[(int64_t)b * gstride + NP] = lsum; }
    int st = 0;
    if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
    if (!ok) st |= PDP_STATUS_PIVOT;
    if (lane == 0 && status) status[b] = st;
#ifdef PDP_PHASE_TIMING   // debug builds (probes/phase_timing.py): cycle stamps of blocks 0 and 700 behind loss[B]
    PDP_STAMP();
    if (lane == 0 && (b == 0 || b == 700)) { long long* o = (long long*)(loss + B) + (b ? 8 : 0); for (int i = 0; i < nst; ++i) o[i] = tstamp[i]; }
    if (lane == 0 && b == 0) { long long* o = (long long*)(loss + B) + 16; for (int i = 0; i < 16; ++i) o[i] = fine[i]; }
    if (lane == 0 && b == 0) { long long* o = (long long*)(loss + B) + 32; for (int i = 0; i < 8; ++i) o[i] = tacc[i]; }
#endif
}

// ------------------------------------------------------------------------------------------------------
// First-order prediction of the optimal trajectory at theta + dtheta - what the auxiliary control system is the derivative OF (PDP.py:582-608: X = dx/dtheta,
// U = du/dtheta, Lambda_t = P_{t+1} X_{t+1} + W_{t+1} = dlambda_{t+1}/dtheta):
//     x_t += X_t dtheta      u_t += U_t dtheta      lam_t += P_{t+1} (X_{t+1} dtheta) + W_{t+1} dtheta
// from the outputs of the gradient unit (dxdp, dudp, riccati).  The starting point of the next OC solve of an IRL loop: one Newton iteration fewer than a
// start from the previous solution (oracle/ipopt_ms.py: predict_start).  16 lanes per stage (lane = row), four stages per wavefront; the rows of a stage are
// contiguous, so a wave reads whole lines.  riccati / lam may be NULL (states and controls only).
// ------------------------------------------------------------------------------------------------------
template <class Mdl>
__global__ void __launch_bounds__(64) oc_predict_kernel(int B, int T, const double* __restrict__ dtheta, int dtb, const double* __restrict__ dxdp,
                                                        const double* __restrict__ dudp, const double* __restrict__ riccati, double* __restrict__ x,
                                                        double* __restrict__ u, double* __restrict__ lam) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, RSZ = oc_riccati_doubles<Mdl>();
    static_assert(NX <= 16 && NU <= 16, "one 16-lane group per stage");
    const int nq = (T + 3) / 4;
    const int b = blockIdx.x / nq, t = (blockIdx.x - b * nq) * 4 + (threadIdx.x >> 4), i = threadIdx.x & 15;
    if (b >= B) return;
    const bool live = t < T;
    const int tc = live ? t : T - 1;                        // (lanes behind the horizon compute stage T - 1 again and store nothing: no divergence around the shuffles)
    double d[NP > 0 ? NP : 1];
#pragma unroll
    for (int j = 0; j < NP; ++j) d[j] = dtheta[(int64_t)b * dtb + j];
    double dx = 0.0;
    if (i < NX) {                                           // node t + 1 (X_0 = 0: x_0 is fixed)
        const double* X = dxdp + (((int64_t)b * (T + 1) + tc + 1) * NX + i) * NP;
#pragma unroll
        for (int j = 0; j < NP; ++j) dx = fma(X[j], d[j], dx);
        if (live) x[((int64_t)b * (T + 1) + tc + 1) * NX + i] += dx;
    }
    if (i < NU) {
        const double* U = dudp + (((int64_t)b * T + tc) * NU + i) * NP;
        double du = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) du = fma(U[j], d[j], du);
        if (live) u[((int64_t)b * T + tc) * NU + i] += du;
    }
    if (riccati && lam) {
        const double* R = riccati + ((int64_t)b * T + tc) * RSZ;
        const int ii = i < NX ? i : 0;
        double dl = 0.0;
#pragma unroll
        for (int j = 0; j < NP; ++j) dl = fma(R[NX * NX + ii * NP + j], d[j], dl);
#pragma unroll
        for (int k = 0; k < NX; ++k) dl = fma(R[ii * NX + k], __shfl(dx, (threadIdx.x & 48) + k, 64), dl);
        if (live && i < NX) lam[((int64_t)b * T + tc) * NX + i] += dl;
    }
}

// the same prediction from the packed fp32 record (PredRec) - the launch of its own used where the runner / evaluator solver kernel, which applies the record
// while loading the point, does not run
template <class Mdl>
__global__ void __launch_bounds__(64) oc_predict_rec_kernel(int B, int T, const double* __restrict__ dtheta, int dtb, const float* __restrict__ rec,
                                                            double* __restrict__ x, double* __restrict__ u, double* __restrict__ lam) {
    using R = PredRec<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    const int nq = (T + 3) / 4;
    const int b = blockIdx.x / nq, t = (blockIdx.x - b * nq) * 4 + (threadIdx.x >> 4), i = threadIdx.x & 15;
    if (b >= B) return;
    const bool live = t < T;
    const int tc = live ? t : T - 1;
    double d[NP > 0 ? NP : 1];
#pragma unroll
    for (int j = 0; j < NP; ++j) d[j] = dtheta[(int64_t)b * dtb + j];
    const float* r = rec + ((int64_t)b * T + tc) * R::SIZE;
    const int ii = i < NX ? i : 0, iu = i < NU ? i : 0;
    double dx = 0.0, du = 0.0, dl = 0.0;
#pragma unroll
    for (int j = 0; j < NP; ++j) dx = fma((double)r[R::X + ii * NP + j], d[j], dx);
#pragma unroll
    for (int j = 0; j < NP; ++j) du = fma((double)r[R::U + iu * NP + j], d[j], du);
    if (i >= NX) dx = 0.0;
    if (live && i < NX) x[((int64_t)b * (T + 1) + tc + 1) * NX + i] += dx;
    if (live && i < NU) u[((int64_t)b * T + tc) * NU + i] += du;
    if (lam) {
#pragma unroll
        for (int j = 0; j < NP; ++j) dl = fma((double)r[R::W + ii * NP + j], d[j], dl);
#pragma unroll
        for (int k = 0; k < NX; ++k) dl = fma((double)r[R::P + R::tri(ii, k)], __shfl(dx, (threadIdx.x & 48) + k, 64), dl);
        if (live && i < NX) lam[((int64_t)b * T + tc) * NX + i] += dl;
    }
}

// ------------------------------------------------------------------------------------------------------
// ControlPlanning (PDP_KIND_CP)
// ------------------------------------------------------------------------------------------------------
template <class Mdl>
__global__ void __launch_bounds__(64) cp_integrate_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0, const double* __restrict__ theta, int tb,
                                    double* __restrict__ x, double* __restrict__ u, double* __restrict__ cost) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* th = theta + (int64_t)b * tb;
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    double xc[NX], xn[NX], uc[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) { xc[i] = x0[(int64_t)b * NX + i]; if (x) x[(int64_t)b * (T + 1) * NX + i] = xc[i]; }
    double J = 0.0;
    for (int t = 0; t < T; ++t) {
        policy_eval<NX, NU>(pol, t, xc, th, uc);
        Mdl::dyn(xc, uc, nullptr, pc, xn);
        J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
        for (int i = 0; i < NU; ++i) if (u) u[((int64_t)b * T + t) * NU + i] = uc[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = xn[i]; if (x) x[((int64_t)b * (T + 1) + t + 1) * NX + i] = xn[i]; }
    }
    if (cost) cost[b] = J + Mdl::final_cost(xc, nullptr, pc);
}

template <class Mdl>
PDP_DEV void fill_static_path(double* dst, int mat, int count) {
    if (!dst) return;
    for (int i = 0; i < count; ++i) {
        int code = Mdl::path_code(mat, i);
        if (code < 0) dst[i] = (code == -1) ? 0.0 : Mdl::path_const(-2 - code);
    }
}

template <class Mdl>
__global__ void __launch_bounds__(64) cp_auxsys_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x, const double* __restrict__ u,
                                 const double* __restrict__ theta, int tb, double* __restrict__ dynF, double* __restrict__ dynG,
                                 double* __restrict__ dUx, double* __restrict__ dUe, double* __restrict__ dcx, double* __restrict__ dcu,
                                 double* __restrict__ dhx) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)B * (T + 1)) return;
    const int b = (int)(g / (T + 1)), t = (int)(g % (T + 1));
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    double xc[NX], uc[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = x[((int64_t)b * (T + 1) + t) * NX + i];
    if (t == T) {
        if (dhx) { double h[NX]; Mdl::dhx(xc, nullptr, pc, h);
#pragma unroll
            for (int i = 0; i < NX; ++i) dhx[(int64_t)b * NX + i] = h[i]; }
        return;
    }
    const int64_t bt = (int64_t)b * T + t;
#pragma unroll
    for (int i = 0; i < NU; ++i) uc[i] = u[bt * NU + i];
    PathDense<Mdl> s;
    s.p[0] = dynF ? dynF + bt * NX * NX : nullptr;
    s.p[1] = dynG ? dynG + bt * NX * NU : nullptr;
    s.p[2] = dcx ? dcx + bt * NX : nullptr;
    s.p[3] = dcu ? dcu + bt * NU : nullptr;
    for (int mat = 0; mat < 4; ++mat) fill_static_path<Mdl>(s.p[mat], mat, Mdl::PATH_ROWS[mat] * Mdl::PATH_COLS[mat]);
    Mdl::eval_path(xc, uc, nullptr, nullptr, pc, s);
    if (dUx && dUe) policy_jacobians<NX, NU>(pol, p, t, xc, theta + (int64_t)b * tb, dUx + bt * NU * NX, dUe + bt * NU * p);
}

// Fused ControlPlanning.step for the Lagrange-polynomial policy: rollout (uniform), then forward sensitivities
// X_{t+1} = F X_t + G Ue_t on MFMA tiles with the per-step matrices staged in LDS; NT tiles of 16 parameters.
// GIVEN (batches with several trajectories per SIMD): trajectory, controls, loss and h_x(x_T) come from cp_poly_rollout_lanes_kernel (one LANE per trajectory, below);
// the kernel keeps only the Jacobian pool and the basis in LDS (rows_given pool rows: sized by the host so that twelve workgroups share a CU) and evaluates its stages
// from global memory - xg [B][T+1][NX], ug [B][T][NU], hxg [B][NX].
template <class Mdl, int NT, bool GIVEN = false>
__global__ void __launch_bounds__(64) cp_step_poly_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0,
                                                           const double* __restrict__ theta, int tb, double* __restrict__ loss,
                                                           double* __restrict__ grad, double* __restrict__ xo, double* __restrict__ uo,
                                                           int rows_given, const double* __restrict__ xg_, const double* __restrict__ ug_, const double* __restrict__ hxg_) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, M = NU;
    const int CH = GIVEN ? rows_given : Mdl::CHUNK;
    constexpr int NC = 1 + Mdl::PATH_NCONST, STRIDE = Mdl::PATH_NVAR | 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* blk = lds;                       // [cpool | pool]
    double* pool = blk + NC;
    double* xs = pool + CH * STRIDE;         // (T+1) x NX     (not GIVEN)
    double* us = xs + (GIVEN ? 0 : (T + 1) * NX);          // T x NU   (not GIVEN)
    double* basis = us + (GIVEN ? 0 : T * NU);             // T x n_pivots
    double* hx = basis + T * pol.n_pivots;   // NX
    [[maybe_unused]] double* dump = hx + NX + 8;              // 64 + max(NX, NU) words nobody reads (see the rollout)
    const int b = blockIdx.x, lane = threadIdx.x, np = pol.n_pivots;
    const int tile0 = blockIdx.y * NT;         // a batch smaller than the machine spreads its parameter tiles over grid.y (rollout repeated)
    [[maybe_unused]] const double* th = theta + (int64_t)b * tb;
    [[maybe_unused]] const double* xg = GIVEN ? xg_ + (int64_t)b * (T + 1) * NX : nullptr;
    [[maybe_unused]] const double* ug = GIVEN ? ug_ + (int64_t)b * T * NU : nullptr;
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    const d4 z = zero4();
    for (int t = lane; t < T; t += 64)
        for (int i = 0; i < np; ++i) basis[t * np + i] = lagrange_basis(pol, i, (double)t);
    if (lane == 0) blk[0] = 0.0;
    for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
    if constexpr (GIVEN) { if (lane < NX) hx[lane] = hxg_[(int64_t)b * NX + lane]; }
    __syncthreads();
    double J = 0.0;
    if constexpr (!GIVEN) {
    // ---- the controls: the Lagrange policy is OPEN-LOOP, u_t = sum_i b_i(t) theta_i depends on t alone - all of them at once, lane = time step, before the rollout.
    // (Evaluated inside the rollout, the np x m parameter loads from global memory sat in the serial chain: ~2.8 k cycles per step, 80 % of this kernel's time.)
    for (int t = lane; t < T; t += 64) {
        double uc[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) uc[j] = 0.0;
        for (int i = 0; i < np; ++i) {
            double bi = basis[t * np + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) uc[j] += bi * th[i * NU + j];
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) us[t * NU + j] = uc[j];
    }
    __syncthreads();
    // ---- rollout (uniform); u_{t+1} is requested from the staging one step ahead
    {
        double xc[NX], xn[NX], uc[NU], un[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) un[j] = us[j];
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int j = 0; j < NU; ++j) { uc[j] = un[j]; un[j] = us[tn * NU + j]; }
            Mdl::dyn(xc, uc, nullptr, pc, xn);
            J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            // x_{t+1} into the LDS staging from lane 0 WITHOUT a conditional block (it would make the wait in front of the next step's LDS
            // reads a full lgkmcnt(0)): every lane stores, the others into a private dump word behind the staging
            {
                double* dx_ = lane == 0 ? xs + (t + 1) * NX : dump + lane;
#pragma unroll
                for (int i = 0; i < NX; ++i) dx_[i] = xn[i];
            }
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        double h[NX];
        Mdl::dhx(xc, nullptr, pc, h);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) hx[i] = h[i];
        }
    }
    __syncthreads();
    if (xo && blockIdx.y == 0) for (int i = lane; i < (T + 1) * NX; i += 64) xo[(int64_t)b * (T + 1) * NX + i] = xs[i];
    if (uo && blockIdx.y == 0) for (int i = lane; i < T * NU; i += 64) uo[(int64_t)b * T * NU + i] = us[i];
    }      // !GIVEN
    // ---- forward sensitivities
    Gather gFT, gGT, gCX, gCU;
    make_gather(gFT, lane, NC, STRIDE, [](int r, int c) { return (r < NX && c < NX) ? Mdl::path_code(0, c * NX + r) : -1; });
    make_gather(gGT, lane, NC, STRIDE, [](int r, int c) { return (r < M && c < NX) ? Mdl::path_code(1, c * NU + r) : -1; });
    make_gather(gCX, lane, NC, STRIDE, [](int r, int c) { return (r < NX) ? Mdl::path_code(2, r) : -1; });
    make_gather(gCU, lane, NC, STRIDE, [](int r, int c) { return (r < M) ? Mdl::path_code(3, r) : -1; });
    d4 X[NT];
    double acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { X[j] = z; acc[j] = 0.0; }
    const int row0 = lane >> 4, col = lane & 15;
    // d pi/d theta = [b_0 I_m ... b_N I_m]: this lane's element of tile j is b_{piv}(t) where piv is fixed (rows < m live in register 0)
    int piv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { const int cidx = 16 * (tile0 + j) + col; piv[j] = (cidx < p && row0 < M && (cidx % NU) == row0) ? cidx / NU : -1; }
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
    for (int c = 0; c < nchunk; ++c) {
        const int t0 = c * ch, cnt = min(ch, T - t0);
        wave_lds_sync();
        if (lane < cnt) {
            const int t = t0 + lane;
            double xc[NX], uc[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = GIVEN ? xg[t * NX + i] : xs[t * NX + i];
#pragma unroll
            for (int i = 0; i < NU; ++i) uc[i] = GIVEN ? ug[t * NU + i] : us[t * NU + i];
            PackedSink s{pool + lane * STRIDE};
            Mdl::eval_path(xc, uc, nullptr, nullptr, pc, s);
        }
        wave_lds_sync();
        // per-step operands are gathered one step ahead (running LDS offsets); two steps per trip, the sensitivity tiles and the
        // prefetched operands alternating between two register sets
        GatherRun rFT = gather_at(gFT, 0, blk), rGT = gather_at(gGT, 0, blk), rCX = gather_at(gCX, 0, blk), rCU = gather_at(gCU, 0, blk);
        struct Ops { d4 FT, GT, CX, CU; };
        auto request = [&](Ops& o) { o.FT = gather_run(rFT, 1); o.GT = gather_run<1>(rGT, 1); o.CX = gather_run(rCX, 1); o.CU = gather_run<1>(rCU, 1); };
        auto step = [&](int tl, const Ops& o, Ops& nx, const d4 (&Xc)[NT], d4 (&Xn)[NT]) {
            const int t = t0 + tl;
            if (tl + 1 < cnt) request(nx);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                d4 Ue = z;
                if (piv[j] >= 0) Ue[0] = basis[t * np + piv[j]];
                acc[j] += o.CX[0] * Xc[j][0] + o.CX[1] * Xc[j][1] + o.CX[2] * Xc[j][2] + o.CX[3] * Xc[j][3] + o.CU[0] * Ue[0];
                d4 Xf = mma_tn(o.FT, Xc[j], z);
                Xn[j] = mma_tn_r0(o.GT, Ue, Xf);
            }
        };
        Ops oa, ob;
        d4 Xb[NT];
        request(oa);
        int tl = 0;
        for (; tl + 1 < cnt; tl += 2) { step(tl, oa, ob, X, Xb); step(tl + 1, ob, oa, Xb, X); }
        if (tl < cnt) {
            step(tl, oa, ob, X, Xb);
#pragma unroll
            for (int j = 0; j < NT; ++j) X[j] = Xb[j];
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc[j] += hx[row] * X[j][r]; }
        double a = sum_over_rowgroups(acc[j]);
        if (lane < 16 && 16 * (tile0 + j) + lane < p) grad[(int64_t)b * p + 16 * (tile0 + j) + lane] = a;
    }
    if constexpr (!GIVEN) { if (lane == 0 && blockIdx.y == 0) loss[b] = J; }
}

// Rollout of ControlPlanning.step with the Lagrange policy, ONE LANE per trajectory (the pre-pass of the GIVEN mode above): u_t = sum_i b_i(t) theta_i, x_{t+1} = f(x_t, u_t),
// J = sum c(x_t, u_t) + h(x_T), h_x(x_T) - the same expressions in the same order as the uniform rollouts of cp_step_poly_kernel / cp_step_poly2_kernel.  The basis is
// shared by the workgroup's 64 trajectories (LDS, [T][np]); each lane's parameters sit in LDS lane-minor ([p][64]: conflict-free) so that no global load is in the serial chain.
template <class Mdl>
__global__ void __launch_bounds__(64) cp_poly_rollout_lanes_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0, const double* __restrict__ theta, int tb,
                                                                    double* __restrict__ loss, double* __restrict__ xw, double* __restrict__ uw, double* __restrict__ hxw) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x, np = pol.n_pivots;
    double* basis = lds;                     // T x np
    double* thl = basis + T * np;            // p x 64
    const int b = blockIdx.x * 64 + lane, bb = b < B ? b : B - 1;
    for (int t = lane; t < T; t += 64)
        for (int i = 0; i < np; ++i) basis[t * np + i] = lagrange_basis(pol, i, (double)t);
    for (int k = 0; k < p; ++k) thl[k * 64 + lane] = theta[(int64_t)bb * tb + k];
    __syncthreads();
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    double xc[NX], xn[NX], uc[NU], J = 0.0;
    double* xb = xw + (int64_t)bb * (T + 1) * NX;
    double* ub = uw + (int64_t)bb * T * NU;
    const bool mine = b < B;
#pragma unroll
    for (int i = 0; i < NX; ++i) { xc[i] = x0[(int64_t)bb * NX + i]; if (mine) xb[i] = xc[i]; }
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int j = 0; j < NU; ++j) uc[j] = 0.0;
        for (int i = 0; i < np; ++i) {
            const double bi = basis[t * np + i];
#pragma unroll
            for (int j = 0; j < NU; ++j) uc[j] += bi * thl[(i * NU + j) * 64 + lane];
        }
        Mdl::dyn(xc, uc, nullptr, pc, xn);
        J += Mdl::path_cost(xc, uc, nullptr, pc);
        if (mine) {
#pragma unroll
            for (int j = 0; j < NU; ++j) ub[t * NU + j] = uc[j];
#pragma unroll
            for (int i = 0; i < NX; ++i) xb[(t + 1) * NX + i] = xn[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = xn[i];
    }
    J += Mdl::final_cost(xc, nullptr, pc);
    double h[NX];
    Mdl::dhx(xc, nullptr, pc, h);
    if (mine) {
        loss[b] = J;
#pragma unroll
        for (int i = 0; i < NX; ++i) hxw[(int64_t)b * NX + i] = h[i];
    }
}

// Fused ControlPlanning.step for ANY policy (tanh MLP up to 8 layers x 32 units, or Lagrange), p <= 512, by the adjoint
// (reverse-mode) sweep.  The reference propagates the n x p sensitivities X_t forward (PDP.py:826-834, O(T n^2 p)) and
// contracts them with c_x, c_u, h_x (871-876); the same gradient is
//     mu_T = h_x ;  v_t = c_u + G_t' mu_{t+1} ;  grad += (d pi/d theta)' v_t ;  mu_t = c_x + F_t' mu_{t+1} + (d pi/d x)' v_t
// i.e. O(T (n^2 + p)) - for the 420-parameter quadrotor policy 100x less arithmetic and no HBM traffic for sensitivities.
// One wavefront per trajectory: lanes = MLP rows / state components / parameter slots (8 parameters per lane in registers);
// activations of the forward pass are kept in LDS for the backward pass; F, G, c_x, c_u come from the lane = time-step
// chunk evaluation like in the other fused kernels.
struct AdjLayout { int theta, xs, acts, actw, zs, ds, mu, v, blk, total; };
// `rows`: time steps per evaluation pass (pool rows, <= 64 lanes); `offload`: the stored hidden activations live in a global
// workspace instead of LDS.  Resident, the kernel's LDS footprint (parameters, trajectory, activations, pool) limits a CU to ~2
// workgroups at p = 420, T = 100; with the activations offloaded and the pool sized to the remainder it fits the 40 KB that let 4
// wavefronts (one per SIMD) share a CU - a batch of 1024 then runs in one wave of workgroups instead of two.
template <class Mdl>
__host__ __device__ inline AdjLayout cp_adjoint_layout(const pdp_policy& pol, int p, int T, bool offload, int rows) {
    AdjLayout L;
    int o = 0;
    L.theta = o; o += p;
    L.xs = o; o += (T + 1) * Mdl::NX;
    L.actw = 0;                                        // hidden activations stored per time step (MLP): sum of hidden widths
    if (pol.kind == PDP_POLICY_MLP) for (int k = 0; k + 1 < pol.n_layers; ++k) L.actw += pol.sizes[k];
    L.acts = o; o += offload ? 0 : T * L.actw;
    L.zs = o; o += 8 * MLP_MAX_WIDTH + 1;              // per-layer inputs z_k of the current step (+ a constant 1.0 for the biases)
    L.ds = o; o += 8 * MLP_MAX_WIDTH;                  // per-layer deltas of the current step
    L.mu = o; o += Mdl::NX;
    L.v = o; o += Mdl::NU + Mdl::NX;                   // v_t, then (d pi/dx)' v_t
    L.blk = o; o += 1 + Mdl::PATH_NCONST + rows * (Mdl::PATH_NVAR | 1);
    L.total = o + 8;
    return L;
}
// host: how the adjoint kernel is laid out for (policy, p, T): resident when that fits 40 KB or no workspace is given, else offloaded
template <class Mdl>
__host__ inline void cp_adjoint_plan(const pdp_policy& pol, int p, int T, int B, int n_cu, bool have_ws, bool& offload, int& rows) {
    const int stride = Mdl::PATH_NVAR | 1;
    rows = 64 * stride * 8 <= 34 * 1024 ? 64 : Mdl::CHUNK;       // as many pool rows as a wavefront has lanes when that fits 34 KB
    offload = false;
    const int resident_bytes = cp_adjoint_layout<Mdl>(pol, p, T, false, rows).total * 8;
    if (!have_ws || pol.kind != PDP_POLICY_MLP || resident_bytes <= 40 * 1024) return;
    if ((int64_t)B <= (int64_t)n_cu * (160 * 1024 / resident_bytes)) return;     // the whole batch is resident at once anyway: keep everything in LDS
    const int fixed = cp_adjoint_layout<Mdl>(pol, p, T, true, 0).total;
    const int fit = (40 * 1024 / 8 - fixed) / stride;
    if (fit < 8) return;                               // not even a small pool fits next to the trajectory: stay resident
    offload = true;
    rows = fit < 64 ? fit : 64;
}

template <class Mdl>
__global__ void __launch_bounds__(64) cp_step_adjoint_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0,
                                                              const double* __restrict__ theta, int tb, double* __restrict__ loss,
                                                              double* __restrict__ grad, double* __restrict__ xo, double* __restrict__ uo,
                                                              double* __restrict__ ws_acts, int CH) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, W = MLP_MAX_WIDTH;
    constexpr int NC = 1 + Mdl::PATH_NCONST, STRIDE = Mdl::PATH_NVAR | 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const bool offload = ws_acts != nullptr;
    const AdjLayout L = cp_adjoint_layout<Mdl>(pol, p, T, offload, CH);
    double *ths = lds + L.theta, *xs = lds + L.xs, *acts = lds + L.acts, *zs = lds + L.zs, *ds = lds + L.ds, *mu = lds + L.mu,
           *vv = lds + L.v, *blk = lds + L.blk, *pool = blk + NC;
    const int b = blockIdx.x, lane = threadIdx.x;
    double* actg = offload ? ws_acts + (int64_t)b * T * L.actw : nullptr;      // [T][actw], each element written and re-read by the same lane
    const bool mlp = pol.kind == PDP_POLICY_MLP;
    const int nl = mlp ? pol.n_layers : 0, np = pol.n_pivots;
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    for (int i = lane; i < p; i += 64) ths[i] = theta[(int64_t)b * tb + i];
    if (lane == 0) { blk[0] = 0.0; zs[8 * W] = 1.0; }
    for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
    // layer tables (uniform): parameter offset, rows, cols, offset of the stored activations
    int loff[8], lrows[8], lcols[8], aoff[8];
    {
        int cols = NX, off = 0, ao = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            loff[k] = off; lcols[k] = cols; lrows[k] = (k < nl) ? pol.sizes[k] : 0; aoff[k] = ao;
            if (k < nl) { off += lrows[k] * cols + lrows[k]; if (k + 1 < nl) ao += lrows[k]; cols = lrows[k]; }
        }
    }
    // per-lane parameter slots q = 0..7 (parameter index lane + 64 q): LDS offsets of the two factors of d cost / d theta_j
    int pr[8], pz[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = lane + 64 * q;
        pr[q] = L.ds; pz[q] = L.blk;                   // blk[0] == 0.0 -> contributes nothing
        if (j < p) {
            if (mlp) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < nl && j >= loff[k] && j < loff[k] + lrows[k] * lcols[k] + lrows[k]) {
                        const int e = j - loff[k], nw = lrows[k] * lcols[k];
                        if (e < nw) { pr[q] = L.ds + k * W + e % lrows[k]; pz[q] = L.zs + k * W + e / lrows[k]; }     // vec_F(A_k): row + col*rows
                        else { pr[q] = L.ds + k * W + (e - nw); pz[q] = L.zs + 8 * W; }                             // bias: factor 1.0
                    }
                }
            } else { pr[q] = L.ds + j % NU; pz[q] = L.zs + j / NU; }       // theta = vcat(U_0..U_N): delta = v[j % m], factor = b_{j / m}(t)
        }
    }
    wave_lds_sync();

    // policy forward at time t from xs[t]: leaves layer inputs in zs, pre-activation deltas untouched; returns u in vv[0..NU)
    auto policy_forward = [&](int t, bool store_acts) {
        if (!mlp) {
            if (lane < np) zs[lane] = lagrange_basis(pol, lane, (double)t);
            wave_lds_sync();
            if (lane < NU) { double u = 0.0; for (int i = 0; i < np; ++i) u += zs[i] * ths[i * NU + lane]; vv[lane] = u; }
            wave_lds_sync();
            return;
        }
        if (lane < NX) zs[lane] = xs[t * NX + lane];
        wave_lds_sync();
        for (int k = 0; k < nl; ++k) {
            const int rows = lrows[k], cols = lcols[k];
            double a = 0.0;
            if (lane < rows) {
                a = ths[loff[k] + rows * cols + lane];
                for (int c = 0; c < cols; ++c) a += ths[loff[k] + lane + c * rows] * zs[k * W + c];
            }
            if (k + 1 < nl) {
                double zk = pdp_tanh(a);
                if (lane < rows) { zs[(k + 1) * W + lane] = zk; if (store_acts) { if (offload) actg[t * L.actw + aoff[k] + lane] = zk; else acts[t * L.actw + aoff[k] + lane] = zk; } }
            } else if (lane < NU) vv[lane] = a;
            wave_lds_sync();
        }
    };

    // ---------------- forward rollout
    double J = 0.0;
    {
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
        wave_lds_sync();
        for (int t = 0; t < T; ++t) {
            policy_forward(t, true);
#pragma unroll
            for (int j = 0; j < NU; ++j) uc[j] = vv[j];
            if (uo && lane < NU) uo[((int64_t)b * T + t) * NU + lane] = vv[lane];
            Mdl::dyn(xc, uc, nullptr, pc, xn);
            J += Mdl::path_cost(xc, uc, nullptr, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[(t + 1) * NX + i] = xn[i];
            }
            wave_lds_sync();
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        double h[NX];
        Mdl::dhx(xc, nullptr, pc, h);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) mu[i] = h[i];
        }
        wave_lds_sync();
    }
    if (xo) for (int i = lane; i < (T + 1) * NX; i += 64) xo[(int64_t)b * (T + 1) * NX + i] = xs[i];

    // ---------------- adjoint sweep
    double gacc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) gacc[q] = 0.0;
    // per-lane pool offsets: column `lane` of F (lane < NX) and of G (lane < NU), entries c_x[lane], c_u[lane]
    int fo[NX], fm[NX], go[NX], gm[NX], cxo = 0, cxm = 0, cuo = 0, cum = 0;
    auto enc = [&](int code, int& o, int& m) { if (code >= 0) { o = NC + code; m = STRIDE; } else { o = (code == -1) ? 0 : 1 + (-2 - code); m = 0; } };
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        enc(lane < NX ? Mdl::path_code(0, k * NX + lane) : -1, fo[k], fm[k]);
        enc(lane < NU ? Mdl::path_code(1, k * NU + lane) : -1, go[k], gm[k]);
    }
    enc(lane < NX ? Mdl::path_code(2, lane) : -1, cxo, cxm);
    enc(lane < NU ? Mdl::path_code(3, lane) : -1, cuo, cum);
    // offloaded activations of step t are requested from the workspace one step ahead (the sweep visits t = T-1 .. 0 in order)
    double apre[8];
    auto request_acts = [&](int t, double (&a)[8]) {
#pragma unroll
        for (int k = 1; k < 8; ++k) a[k] = (k < nl && lane < lrows[k - 1]) ? actg[t * L.actw + aoff[k - 1] + lane] : 0.0;
    };
    if (offload) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); request_acts(T - 1, apre); }
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
    for (int c = nchunk - 1; c >= 0; --c) {
        const int t0 = c * ch, cnt = min(ch, T - t0);
        wave_lds_sync();
        if (lane < cnt) {
            const int t = t0 + lane;
            double xc[NX], uc[NU];
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xs[t * NX + i];
            // u_t is re-derived from the stored trajectory through the policy only for the MLP (cheap for the polynomial)
            if (!mlp) {
#pragma unroll
                for (int j = 0; j < NU; ++j) { double u = 0.0; for (int i = 0; i < np; ++i) u += lagrange_basis(pol, i, (double)t) * ths[i * NU + j]; uc[j] = u; }
            } else {
                policy_eval<NX, NU>(pol, t, xc, ths, uc);
            }
            PackedSink sk{pool + lane * STRIDE};
            Mdl::eval_path(xc, uc, nullptr, nullptr, pc, sk);
        }
        wave_lds_sync();
        for (int tl = cnt - 1; tl >= 0; --tl) {
            const int t = t0 + tl;
            // v = c_u + G' mu
            if (lane < NU) {
                double a = blk[cuo + tl * cum];
#pragma unroll
                for (int k = 0; k < NX; ++k) a += blk[go[k] + tl * gm[k]] * mu[k];
                vv[lane] = a;
            }
            wave_lds_sync();
            // policy backward: deltas per layer into ds, layer inputs into zs
            if (!mlp) {
                if (lane < np) zs[lane] = lagrange_basis(pol, lane, (double)t);
                if (lane < NU) ds[lane] = vv[lane];
                if (lane < NX) vv[NU + lane] = 0.0;                   // d pi/dx = 0
                wave_lds_sync();
            } else {
                if (lane < NX) zs[lane] = xs[t * NX + lane];
                if (offload) {
                    double anext[8];
                    if (t > 0) request_acts(t - 1, anext);
#pragma unroll
                    for (int k = 1; k < 8; ++k) if (k < nl && lane < lrows[k - 1]) zs[k * W + lane] = apre[k];
#pragma unroll
                    for (int k = 1; k < 8; ++k) apre[k] = (t > 0) ? anext[k] : 0.0;
                } else {
                    for (int k = 1; k < nl; ++k) if (lane < lrows[k - 1]) zs[k * W + lane] = acts[t * L.actw + aoff[k - 1] + lane];
                }
                if (lane < NU) ds[(nl - 1) * W + lane] = vv[lane];
                wave_lds_sync();
                for (int k = nl - 1; k >= 0; --k) {
                    const int rows = lrows[k], cols = lcols[k];
                    double a = 0.0;
                    if (lane < cols) for (int r = 0; r < rows; ++r) a += ths[loff[k] + r + lane * rows] * ds[k * W + r];      // A_k' delta_k
                    if (k > 0) { if (lane < cols) { double zk = zs[k * W + lane]; ds[(k - 1) * W + lane] = a * (1.0 - zk * zk); } }
                    else if (lane < NX) vv[NU + lane] = a;                                                                     // (d pi/dx)' v
                    wave_lds_sync();
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) gacc[q] += lds[pr[q]] * lds[pz[q]];
            // mu_t = c_x + F' mu_{t+1} + (d pi/dx)' v
            double m_new = 0.0;
            if (lane < NX) {
                m_new = blk[cxo + tl * cxm] + vv[NU + lane];
#pragma unroll
                for (int k = 0; k < NX; ++k) m_new += blk[fo[k] + tl * fm[k]] * mu[k];
            }
            wave_lds_sync();
            if (lane < NX) mu[lane] = m_new;
            wave_lds_sync();
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int j = lane + 64 * q; if (j < p) grad[(int64_t)b * p + j] = gacc[q]; }
    if (lane == 0) loss[b] = J;
}

// ------------------------------------------------------------------------------------------------------
// SysID (PDP_KIND_SYSID)
// ------------------------------------------------------------------------------------------------------
template <class Mdl>
__global__ void __launch_bounds__(64) sysid_integrate_kernel(int B, int T, const double* __restrict__ x0, int x0_stride, const double* __restrict__ u,
                                                              const double* __restrict__ theta, int tb, double* __restrict__ x) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double th[Mdl::NP > 0 ? Mdl::NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    // lane = trajectory.  The controls of a lane are requested TWO steps ahead of their use: read inside the step they would put a round trip to memory (every lane in
    // its own cache line) into the serial chain of each of the T steps.  x0_stride: NX (x0 [B][NX]) or (T + 1) NX (the initial states are the first rows of a
    // trajectory array: SysID.step's batch_states[i][0], PDP.py:1269)
    double xc[NX], xn[NX], uc[NU], u1[NU], u2[NU];
    const double* ub = u + (int64_t)b * T * NU;
#pragma unroll
    for (int i = 0; i < NX; ++i) { xc[i] = x0[(int64_t)b * x0_stride + i]; x[(int64_t)b * (T + 1) * NX + i] = xc[i]; }
#pragma unroll
    for (int i = 0; i < NU; ++i) { u1[i] = ub[i]; u2[i] = ub[(T > 1 ? 1 : 0) * NU + i]; }
    for (int t = 0; t < T; ++t) {
        const int tn = t + 2 < T ? t + 2 : T - 1;
#pragma unroll
        for (int i = 0; i < NU; ++i) { uc[i] = u1[i]; u1[i] = u2[i]; u2[i] = ub[tn * NU + i]; }
        Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
        for (int i = 0; i < NX; ++i) { xc[i] = xn[i]; x[((int64_t)b * (T + 1) + t + 1) * NX + i] = xn[i]; }
    }
}

template <class Mdl>
__global__ void __launch_bounds__(64) sysid_auxsys_kernel(int B, int T, const double* __restrict__ x, const double* __restrict__ u, const double* __restrict__ theta,
                                    int tb, double* __restrict__ dynF, double* __restrict__ dynE) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)B * T) return;
    const int b = (int)(g / T);
    double th[NP > 0 ? NP : 1];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    const int t = (int)(g % T);
    double xc[NX], uc[NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) xc[i] = x[((int64_t)b * (T + 1) + t) * NX + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) uc[i] = u[g * NU + i];
    PathDense<Mdl> s;
    s.p[0] = dynF ? dynF + g * NX * NX : nullptr;
    s.p[1] = dynE ? dynE + g * NX * NP : nullptr;
    s.p[2] = s.p[3] = nullptr;
    fill_static_path<Mdl>(s.p[0], 0, NX * NX);
    fill_static_path<Mdl>(s.p[1], 1, NX * NP);
    Mdl::eval_path(xc, uc, nullptr, th, pc, s);
}

// doubles of LDS per trajectory of the fused SysID.step kernels (sysid_step_kernel below, sysid_step2_kernel in pdp_cp_pair_kernels.h):
// [cpool | pool CH rows | x (T+1) x NX | dlT NX + 1 | u T x NU | dump 64 + NX | hand-over counter, padding]
template <class Mdl>
__host__ __device__ inline int sysid_slice(int T, int rows = Mdl::CHUNK, bool given = false) {      // given: the trajectory and the controls stay in global memory (no staging)
    const int n = 1 + Mdl::PATH_NCONST + rows * ((Mdl::PATH_NVAR + Mdl::NX) | 1) + (given ? 0 : (T + 1) * Mdl::NX) + Mdl::NX + 1 + (given ? 0 : T * Mdl::NU) + 64 + Mdl::NX + 8;
    return (n + 1) & ~1;
}
// pool rows of the given-trajectory mode: what lets `wgs` workgroups (wavefronts) share a CU's 160 KB.  12 = three wavefronts per SIMD, what the 162 VGPRs of that
// instantiation allow (forced to 128 registers for four waves it spills 28); measured 8 / 12 / 16: 0.316 / 0.293 / 0.298 ms at B = 8192 (profiles/r04_rollout_prepass.txt)
template <class Mdl>
__host__ inline int sysid_rows_given(int T, int wgs) {
    const int stride = (Mdl::PATH_NVAR + Mdl::NX) | 1;
    const int fit = (160 * 1024 / 8 / wgs - 64 - sysid_slice<Mdl>(T, 0, true)) / stride;
    return fit >= 4 ? (fit < Mdl::CHUNK ? fit : Mdl::CHUNK) : (Mdl::CHUNK < 4 ? Mdl::CHUNK : 4);
}
// Pool rows (stages per lane-parallel Jacobian pass) of sysid_step_kernel.  The generated CHUNK (a 17 KB pool for the quadrotor) makes a workgroup 31 KB: five
// wavefronts per CU - fine while there is at most one trajectory per SIMD.  A batch with several trajectories per SIMD (C5's total of 8192 on one GPU: eight rounds)
// is served better by TWO resident waves per SIMD that fill each other's latency gaps (two waves take 1.4x the time of one): the pool is
// cut to what lets eight workgroups share the CU's 160 KB.
template <class Mdl>
__host__ inline int sysid_rows(int B, int T, int cus) {
    if (B <= 4 * cus) return Mdl::CHUNK;
    const int stride = (Mdl::PATH_NVAR + Mdl::NX) | 1;
    const int fit = (160 * 1024 / 8 / 8 - 64 - sysid_slice<Mdl>(T, 0)) / stride;      // (64 doubles of slack for the allocation granularity)
    return fit >= 4 ? (fit < Mdl::CHUNK ? fit : Mdl::CHUNK) : Mdl::CHUNK;
}
// Fused SysID.step per trajectory: rollout (uniform, x kept in LDS) then X_{t+1} = F X_t + E on MFMA tiles.
template <class Mdl, int NT, bool GIVEN = false>
__global__ void __launch_bounds__(64) sysid_step_kernel(int B, int T, const double* __restrict__ u, const double* __restrict__ xobs,
                                                         const double* __restrict__ theta, int tb, double* __restrict__ loss, double* __restrict__ grad, int CH,
                                                         const double* __restrict__ xgiven_) {
    const double* __restrict__ xgiven = GIVEN ? xgiven_ : nullptr;      // (a template parameter: each instantiation keeps its own register allocation)
    // xgiven [B][T+1][NX] (GIVEN): the trajectory, rolled out beforehand by sysid_integrate_kernel with ONE LANE per trajectory - the mode for batches with several
    // trajectories per SIMD: the rollout below runs one trajectory on all 64 lanes (the same value in every lane), which is the right thing while the SIMD has nothing
    // else to do and 63/64 wasted once other trajectories wait for it (C5a's 8192 on one GPU: 8 rounds of a 26 us rollout against one 21 us pass for all of them)
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    constexpr int NC = 1 + Mdl::PATH_NCONST, DLX = Mdl::PATH_NVAR, STRIDE = (Mdl::PATH_NVAR + NX) | 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* blk = lds;
    double* pool = blk + NC;
    double* xs = pool + CH * STRIDE;         // (T+1) x NX   (not in the given-trajectory mode: stages are evaluated from global memory there)
    double* dlT = xs + (xgiven ? 0 : (T + 1) * NX);         // NX
    double* us = dlT + NX + 1;               // T x NU: the given controls, staged once with coalesced loads (rollout mode only)
    double* dump = us + (xgiven ? 0 : T * NU);              // 64 + NX words nobody reads (see the rollout)
    const int b = blockIdx.x, lane = threadIdx.x;
    const d4 z = zero4();
    double th[NP];
    load_theta<Mdl>(theta, b, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    const double* ub = u + (int64_t)b * T * NU;
    const double* ob = xobs + (int64_t)b * (T + 1) * NX;
    if (lane == 0) blk[0] = 0.0;
    for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
    const double* xg = xgiven ? xgiven + (int64_t)b * (T + 1) * NX : nullptr;
    if (!xgiven) for (int q = lane; q < T * NU; q += 64) us[q] = ub[q];
    __syncthreads();
    if constexpr (!GIVEN) {
        // rollout (uniform).  u_t comes from the LDS staging, requested one step ahead (read from global memory inside the loop every step waited for a round
        // trip to memory); x_{t+1} goes to the staging from lane 0 without a conditional block (cp_step_poly_kernel)
        double xc[NX], xn[NX], uc[NU], un[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = ob[i];                       // ini_state = batch_states[i][0] (PDP.py:1269)
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
#pragma unroll
        for (int i = 0; i < NU; ++i) un[i] = us[i];
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
            Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            double* dx_ = lane == 0 ? xs + (t + 1) * NX : dump + lane;
#pragma unroll
            for (int i = 0; i < NX; ++i) dx_[i] = xn[i];
        }
    }
    __syncthreads();
    Gather gFT, gDX, gE[NT];
    make_gather(gFT, lane, NC, STRIDE, [](int r, int c) { return (r < NX && c < NX) ? Mdl::path_code(0, c * NX + r) : -1; });
    make_gather(gDX, lane, NC, STRIDE, [](int r, int c) { return (r < NX) ? DLX + r : -1; });
#pragma unroll
    for (int j = 0; j < NT; ++j)
        make_gather(gE[j], lane, NC, STRIDE, [j](int r, int c) { return (r < NX && 16 * j + c < NP) ? Mdl::path_code(1, r * NP + 16 * j + c) : -1; });
    d4 X[NT];
    double acc[NT], lsum = 0.0;
#pragma unroll
    for (int j = 0; j < NT; ++j) { X[j] = z; acc[j] = 0.0; }
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;      // chunks of equal length
    for (int c = 0; c < nchunk; ++c) {
        const int t0 = c * ch, cnt = min(ch, T - t0);
        __syncthreads();
        if (lane < cnt) {
            const int t = t0 + lane;
            double xc[NX], uc[NU];
            double* row = pool + lane * STRIDE;
#pragma unroll
            for (int i = 0; i < NX; ++i) { xc[i] = xg ? xg[t * NX + i] : xs[t * NX + i]; double d = xc[i] - ob[t * NX + i]; row[DLX + i] = d; lsum += d * d; }
#pragma unroll
            for (int i = 0; i < NU; ++i) uc[i] = xg ? ub[t * NU + i] : us[t * NU + i];
            PackedSink s{row};
            Mdl::eval_path(xc, uc, nullptr, th, pc, s);
        }
        __syncthreads();
        for (int tl = 0; tl < cnt; ++tl) {
            d4 FT = gather_tile(blk, gFT, tl);
            d4 DX = gather_tile(blk, gDX, tl);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                d4 E = gather_tile(blk, gE[j], tl);
                acc[j] += DX[0] * X[j][0] + DX[1] * X[j][1] + DX[2] * X[j][2] + DX[3] * X[j][3];
                X[j] = mma_tn(FT, X[j], E);
            }
        }
    }
    __syncthreads();
    if (lane < NX) { double d = (xg ? xg[T * NX + lane] : xs[T * NX + lane]) - ob[T * NX + lane]; dlT[lane] = d; lsum += d * d; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc[j] += dlT[row] * X[j][r]; }
        double a = sum_over_rowgroups(acc[j]);
        if (lane < 16 && 16 * j + lane < NP) grad[(int64_t)b * NP + 16 * j + lane] = a;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) loss[b] = lsum;
}

}  // namespace pdp
