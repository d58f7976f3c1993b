// pdp_ocsolve_kernels.h - the multiple-shooting optimal-control solve, one persistent wavefront per trajectory.
//
// Stands where the reference's OCSys.ocSolver hands its NLP to IPOPT (PDP/PDP.py:121-220):
//     min  sum_t c(x_t,u_t) + h(x_T)   over  x_1..x_T, u_0..u_{T-1}     s.t.  f(x_t,u_t) - x_{t+1} = 0,   x_0 fixed,
// all-zero initial guess (PDP.py:155,166), no active bounds (+-1e20).  The iteration is IPOPT's for that case - Waechter &
// Biegler, Math. Program. 106 (2006): primal-dual Newton step on the KKT system, inertia correction W + dw I (Algorithm IC),
// filter line search on (theta = |c|_1, phi = f) with the switching / Armijo / sufficient-decrease rules (Algorithm A), the
// multipliers moving with the primal step length, least-squares initial multipliers.  oracle/ipopt_ms.py is the CPU
// restatement this kernel is tested against, pinned on the optima the reference stored.
//
// The Newton step is an LQ problem with affine terms - defects c_t in the dynamics, Lagrangian gradients in the cost - so it is
// solved by the SAME backward Riccati / forward rollout on 16x16 fp64 MFMA tiles as LQR.lqrSolver (pdp_riccati.h) with one
// "parameter" column:   Y2 = [G | c_t]   HX2 = [Hxu | grad_x L]   HU2 = [Huu | grad_u L]   W2 = [0 | W],
// and the KKT matrix has the right inertia iff every Quu_t of the sweep is positive definite.  Per iteration:
//   backward sweep in chunks: lane = stage evaluates F, G, Hxx, Hxu, Huu at (x_t, u_t, lambda_{t+1}), the defect and the Lagrangian
//       gradients into the LDS pool (nothing of this touches HBM), then the Riccati steps gather their tiles from the pool;
//   forward pass: (dx, du) and the multiplier step dlam_t = P_{t+1} dx_{t+1} + W_{t+1}   (PDP.py:604);
//   filter line search: every trial point is evaluated with lane = stage (no serial rollout anywhere in this solver);
// everything data dependent - inertia retries, step length, convergence - is decided inside the wavefront: no host round trip,
// no batch-wide synchronisation, a trajectory leaves the GPU when ITS iteration has converged.
// Not implemented: IPOPT's second-order correction (changes iteration counts only) and restoration phase (the trajectory is
// returned with PDP_MS_RESTORATION set and the caller falls back to the single-shooting solver, pdp_oc_solve_batched).
#pragma once
#include "pdp_model_kernels.h"

namespace pdp {

template <class Mdl>
struct MsLayout {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = Mdl::MS_CHUNK;
    static constexpr int NS = Mdl::SOL_NVAR, NF = Mdl::SOLF_NVAR;
    static constexpr int NC0 = Mdl::SOL_NCONST > Mdl::SOLF_NCONST ? Mdl::SOL_NCONST : Mdl::SOLF_NCONST;
    static constexpr int NC = 1 + (NC0 > Mdl::FIN_NCONST ? NC0 : Mdl::FIN_NCONST);
    // backward pool row: [sol entries | defect c_t (NX) | grad_x L (NX) | grad_u L (NU)], forward: [solf entries | defect (NX)]
    static constexpr int C0 = NS, RX = NS + NX, RU = NS + 2 * NX;
    static constexpr int BSTRIDE = (NS + 2 * NX + NU) | 1;
    static constexpr int FC0 = NF;
    static constexpr int FSTRIDE = (NF + NX) | 1;
    static constexpr int ROWMAX = BSTRIDE > FSTRIDE ? BSTRIDE : FSTRIDE;
    static constexpr int POOL = CH * ROWMAX > Mdl::FIN_NVAR + 1 ? CH * ROWMAX : Mdl::FIN_NVAR + 1;
    static constexpr int GSZ = NX * NU + NU + 1;                   // per stage: K [NU x NX] | k [NU] | zero sink
    static constexpr int PWSZ = NX * NX + NX + 1;                  // per stage: P_{t+1} [NX x NX] | W_{t+1} [NX] | zero sink
    static constexpr int LDS_DOUBLES = RICCATI_SCRATCH + NC + POOL + NX + NP + Mdl::NPC + 8;
    // workspace per trajectory (doubles): dx | du | dlam | defects | grad_x L | grad_u L | gains | P,W | filter (theta, phi per iteration)
    __host__ __device__ static constexpr int64_t ws_doubles(int T, int max_iter) {
        return (int64_t)(T + 1) * NX + (int64_t)T * NU + (int64_t)T * NX + (int64_t)T * NX + (int64_t)(T + 1) * NX + (int64_t)T * NU +
               (int64_t)T * GSZ + (int64_t)T * PWSZ + 2 * (int64_t)(max_iter + 1);
    }
};

// store column `col` of a tile (rows < R) to dst[row]
PDP_DEV void store_tile_column(double* __restrict__ dst, const d4 v, int R, int col, int lane) {
    if (tile_col(lane) != col) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = tile_row(lane, r); if (row < R) dst[row] = v[r]; }
}
// the same through a BUFFER store: voff[r] = byte offset of this lane's row inside the column, or 0x80000000 where the lane holds no element of it - the
// buffer's range check drops those lanes in hardware, so there is no predicated block (mask, branch, 64-bit address arithmetic) per register
typedef unsigned pdp_u2 __attribute__((ext_vector_type(2)));
struct ColStore { unsigned voff[4]; };
PDP_DEV ColStore make_col_store(int R, int col, int lane) {
    ColStore c;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int row = tile_row(lane, r); c.voff[r] = (tile_col(lane) == col && row < R) ? 8u * (unsigned)row : 0x80000000u; }
    return c;
}
template <int NR = 4, class RS>
PDP_DEV void store_tile_column_buf(RS rs, unsigned soff, const ColStore& c, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double x = v[r];
        pdp_u2 w;
        w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x);
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, c.voff[r], soff, 0);
    }
}
PDP_DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

template <class Mdl>
__global__ void __launch_bounds__(64) oc_solve_ms_kernel(int B, int T, pdp_oc_ms_opts op, const double* __restrict__ x0, const double* __restrict__ theta,
                                                          int tb, double* __restrict__ x, double* __restrict__ u, double* __restrict__ lam,
                                                          double* __restrict__ cost, double* __restrict__ resid, int32_t* __restrict__ converged,
                                                          int32_t* __restrict__ iters, int32_t* __restrict__ status, double* __restrict__ gains_out,
                                                          double* __restrict__ iter_log, double* __restrict__ ws) {
    using L = MsLayout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = L::CH, M = NU;
    constexpr int GSZ = L::GSZ, PWSZ = L::PWSZ;
    // SMALL (n <= 4): the whole recursion on the rows-0..3 register of each tile, every product ONE 4-block MFMA (pdp_riccati_small.h);
    // left operands are gathered / loaded in "rep" form (the 4 x 4 block replicated in the four column blocks)
    constexpr bool SMALL = NX <= 4;
    constexpr int NRT = SMALL ? 1 : 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* scratch = lds;                              // RICCATI_SCRATCH
    double* blk = lds + RICCATI_SCRATCH;                // [constants (NC) | pool]
    double* pool = blk + L::NC;
    double* dlT = pool + L::POOL;                       // terminal gradient (NX)
    double* par = dlT + NX;                             // [theta (NP) | pc (NPC)]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int tlane = small_transpose_lane(lane);
    const d4 z = zero4();
    {
        double th0[NP > 0 ? NP : 1], pc0[Mdl::NPC];
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
#define PDP_MS_PAR()                                                      \
    double th[NP > 0 ? NP : 1], pc[Mdl::NPC];                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) th[i_] = par[i_];  \
    _Pragma("unroll") for (int i_ = 0; i_ < Mdl::NPC; ++i_) pc[i_] = par[NP + i_]
#ifdef PDP_MS_TIMING      // timing builds (probes/ms_phase_timing.py): cycles per phase and iteration in the iteration log instead of IPOPT's columns
    long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm0 = 0, tmi = 0;
#define MS_T0() tm0 = __builtin_readcyclecounter()
#define MS_T1(k) do { const long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tm0; tm0 = now_; } while (0)
#else
#define MS_T0()
#define MS_T1(k)
#endif
    double* xb = x + (int64_t)b * (T + 1) * NX;
    double* ub = u + (int64_t)b * T * NU;
    double* lb = lam + (int64_t)b * T * NX;
    double* w0 = ws + (int64_t)b * L::ws_doubles(T, op.max_iter);
    double* dxb = w0;                                   // (T+1) x NX
    double* dub = dxb + (int64_t)(T + 1) * NX;          // T x NU
    double* dlb = dub + (int64_t)T * NU;                // T x NX
    double* cst = dlb + (int64_t)T * NX;                // defects c_t, T x NX
    double* rxs = cst + (int64_t)T * NX;                // grad_x L, (T+1) x NX  (row 0 unused: x_0 is fixed)
    double* rus = rxs + (int64_t)(T + 1) * NX;          // grad_u L, T x NU
    double* gw = rus + (int64_t)T * NU;                 // gains, T x GSZ
    double* pw = gw + (int64_t)T * GSZ;                 // P_{t+1}, W_{t+1}, T x PWSZ
    double* fth = pw + (int64_t)T * PWSZ;               // filter: theta entries (at most one per iteration) ...
    double* fph = fth + (op.max_iter + 1);              //         ... and phi entries

    // ---- starting point: the caller's (x, u, lambda) [PDP_MS_WARM], or IPOPT's: w0 = 0 (PDP.py:155,166), x_0 = ini_state ------
    const bool warm = (op.flags & PDP_MS_WARM) != 0;
    if (!warm) {
        for (int i = lane; i < (T + 1) * NX; i += 64) xb[i] = i < NX ? x0[(int64_t)b * NX + i] : 0.0;
        for (int i = lane; i < T * NU; i += 64) ub[i] = 0.0;
        for (int i = lane; i < T * NX; i += 64) lb[i] = 0.0;
    } else if (lane < NX) xb[lane] = x0[(int64_t)b * NX + lane];
    for (int i = lane; i < NX; i += 64) { dxb[i] = 0.0; rxs[i] = 0.0; }
    __threadfence_block();
    wave_lds_sync();

    // ---- loop-invariant gather / store maps ---------------------------------------------------------------------------------------
    auto codeS = [](int mat, int i) { return Mdl::sol_code(mat, i); };       // 0 F, 1 G, 2 Hxx, 3 Hxu, 4 Huu
    Gather gF, gY, gHxx, gHX, gHU, gGr, gHux;
    make_gather(gF, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(0, r * NX + (c & 3)) : -1)
                                                                              : ((r < NX && c < NX) ? codeS(0, r * NX + c) : -1); });
    make_gather(gY, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(1, r * NU + c) : (c == M ? L::C0 + r : -1)); });
    make_gather(gGr, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeS(1, r * NU + (c & 3)) : -1; });
    make_gather(gHux, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? codeS(3, (c & 3) * NU + r) : -1)
                                                                                : ((r < M && c < NX) ? codeS(3, c * NU + r) : -1); });
    make_gather(gHxx, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(2, r * NX + (c & 3)) : -1)
                                                                                : ((r < NX && c < NX) ? codeS(2, r * NX + c) : -1); });
    make_gather(gHX, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(3, r * NU + c) : (c == M ? L::RX + r : -1)); });
    make_gather(gHU, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return r >= M ? -1 : (c < M ? codeS(4, r * NU + c) : (c == M ? L::RU + r : -1)); });
    const TileMapBytes mK = make_tile_map_sink(NU, NX, NX, 0, 0, lane, GSZ - 1), mIK = make_tile_map_sink(NU, 1, 1, 0, M, lane, NU),
                       mP = make_tile_map_sink(NX, NX, NX, 0, 0, lane, PWSZ - 1), mW = make_tile_map_sink(NX, 1, 1, 0, M, lane, NX),
                       mPld = SMALL ? to_bytes_sink(make_rep4_map(NX, NX, NX, lane), PWSZ - 1) : mP,      // P read back for the multiplier step (rep form when SMALL)
                       mKT = to_bytes_sink(make_rep4_map_transposed(NX, NU, NX, lane), GSZ - 1);
    const int col = tile_col(lane);
    // per-lane tile masks: diagonal of the n x n / m x m blocks, control columns, the affine column
    d4 dgN, dgM0 = z;
#pragma unroll
    for (int r = 0; r < 4; ++r) dgN[r] = SMALL ? ((r == 0 && (lane >> 4) == (col & 3) && (col & 3) < NX) ? 1.0 : 0.0) : ((tile_row(lane, r) == col && col < NX) ? 1.0 : 0.0);
    dgM0[0] = ((lane >> 4) == col && col < M) ? 1.0 : 0.0;

    // ---- per-sweep reductions over the stages (complete after a sweep that was not aborted) ------------------------------------------
    double f_cur = 0.0, th_cur = 0.0, inf_pr = 0.0, inf_du = 0.0, zmax = 0.0, lmax = 0.0, lamc = 0.0;
    bool finite = true;

    // Backward sweep with Hessian scale hs (1; 0 = least-squares multiplier estimate: W = I, no defects) and shift dw.
    // Returns true when every Quu was positive definite; aborts at the first one that is not.
    auto backward = [&](double hs, double dw) -> bool {
        const double sU = col < M ? hs : 1.0, sC = col == M ? hs : 1.0;      // scale of the Hessian columns / of the defect column
        double a_f = 0.0, a_th = 0.0, a_pr = 0.0, a_du = 0.0, a_z = 0.0, a_l = 0.0, a_lc = 0.0;
        bool fin = true, pdall = true, ok = true;
        // terminal stage: P = hs hxx + dw I, W = h_x(x_T) - lambda_T
        MS_T0();
        wave_lds_sync();
        if (lane == 0) blk[0] = 0.0;
        for (int i = lane; i < Mdl::FIN_NCONST; i += 64) blk[1 + i] = Mdl::fin_const(i);
        if (lane == 0) {
            PDP_MS_PAR();
            double xT[NX], hx[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xT[i] = xb[T * NX + i];
            PackedSink s{pool};
            Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
            Mdl::dhx(xT, th, pc, hx);
            a_f += Mdl::final_cost(xT, th, pc);
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double g = hx[i] - lb[(T - 1) * NX + i];
                dlT[i] = g; rxs[T * NX + i] = g;
                a_du = fmax(a_du, fabs(g)); a_z = fmax(a_z, fabs(xT[i]));
                fin = fin && fabs(g) <= 1.7e308;
            }
        }
        wave_lds_sync();
        d4 P, W2 = z;
        {
            Gather gP;
            make_gather(gP, lane, L::NC, 0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::fin_code(0, r * NX + (c & 3)) : -1)
                                                                             : ((r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1); });
            P = gather_tile(blk, gP, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile_row(lane, r);
                P[r] = hs * P[r] + dw * dgN[r];
                if (col == M && row < NX) W2[r] = dlT[row];
            }
        }
        wave_lds_sync();
        for (int i = lane; i < Mdl::SOL_NCONST; i += 64) blk[1 + i] = Mdl::sol_const(i);
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;
        for (int c = nchunk - 1; c >= 0 && pdall; --c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            wave_lds_sync();
            MS_T1(1);
            if (lane < cnt) {                       // lane = stage: KKT matrices, defect, Lagrangian gradients at (x_t, u_t, lambda_{t+1})
                PDP_MS_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU], lc[NX], v[NX];
                double* row = pool + lane * L::BSTRIDE;
#pragma unroll
                for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; lc[i] = lb[t * NX + i]; a_z = fmax(a_z, fabs(xc[i])); a_l = fmax(a_l, fabs(lc[i])); }
#pragma unroll
                for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; a_z = fmax(a_z, fabs(uc[i])); }
                PackedSink s{row};
                Mdl::eval_sol(xc, uc, lc, th, pc, s);
                Mdl::dyn(xc, uc, th, pc, v);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double ci = v[i] - xb[(t + 1) * NX + i];
                    row[L::C0 + i] = ci; cst[t * NX + i] = ci;
                    a_th += fabs(ci); a_pr = fmax(a_pr, fabs(ci)); a_lc += lc[i] * ci;
                    fin = fin && fabs(ci) <= 1.7e308;
                }
                Mdl::costate_step(xc, uc, lc, th, pc, v);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double g = t > 0 ? v[i] - lb[(t - 1) * NX + i] : 0.0;       // x_0 is fixed: no stationarity row
                    row[L::RX + i] = g; rxs[t * NX + i] = g;
                    a_du = fmax(a_du, fabs(g));
                    fin = fin && fabs(g) <= 1.7e308;
                }
                double hu[NU];
                Mdl::dHu(xc, uc, lc, th, pc, hu);
#pragma unroll
                for (int i = 0; i < NU; ++i) { row[L::RU + i] = hu[i]; rus[t * NU + i] = hu[i]; a_du = fmax(a_du, fabs(hu[i])); fin = fin && fabs(hu[i]) <= 1.7e308; }
                a_f += Mdl::path_cost(xc, uc, th, pc);
            }
            wave_lds_sync();
            MS_T1(2);
            GatherRun rF = gather_at(gF, cnt - 1, blk), rY = gather_at(gY, cnt - 1, blk), rHxx = gather_at(gHxx, cnt - 1, blk), rHX = gather_at(gHX, cnt - 1, blk),
                      rHU = gather_at(gHU, cnt - 1, blk), rGr = gather_at(gGr, cnt - 1, blk), rHux = gather_at(gHux, cnt - 1, blk);
            d4 Fa = gather_run<NRT>(rF, -1), Ya = gather_run<NRT>(rY, -1), Fb = z, Yb = z;
            auto bstep = [&](int tl, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {
                const int t = t0 + tl;
                d4 Hxx = gather_run<NRT>(rHxx, -1), HX2 = gather_run<NRT>(rHX, -1), HU2 = gather_run<1>(rHU, -1), Grep = gather_run<NRT>(rGr, -1), Hux = gather_run<1>(rHux, -1);
                if (tl > 0) { Fn = gather_run<NRT>(rF, -1); Yn = gather_run<NRT>(rY, -1); }      // (no prefetch below the first row of the pool)
#pragma unroll
                for (int r = 0; r < 4; ++r) { Hxx[r] = hs * Hxx[r] + dw * dgN[r]; HX2[r] *= sU; }
                HU2[0] = sU * HU2[0] + dw * dgM0[0];
                const d4 Ys = Yc * sC;
                const double Hux0 = hs * Hux[0];
                // P_{t+1}, W_{t+1}: the multiplier step of the forward pass needs them (dlam_t = P_{t+1} dx_{t+1} + W_{t+1})
                if constexpr (SMALL) {
                    d4 Pst = z;
                    Pst[0] = (lane & 12) == 0 ? P[0] : 0.0;      // P is in rep form: only its first column block goes to the workspace (the zero sink stays zero)
                    store_all<1>(pw + t * PWSZ, mP, Pst);
                    store_all<1>(pw + t * PWSZ + NX * NX, mW, W2);
                    SmallGains g;
                    double Pr = P[0], Wr = W2[0];
                    ok = riccati_small_backward<M, true>(Pr, Wr, Fc[0], Ys[0], Grep[0], Hxx[0], HX2[0], HU2[0], Hux0, lane, tlane, 1, g) && ok;
                    P[0] = Pr; W2[0] = Wr;
                    pdall = pdall && g.pd;
                    d4 Kt = z, IKt = z;
                    Kt[0] = (lane & 12) == 0 ? g.K : 0.0;
                    IKt[0] = g.IK;
                    store_all<1>(gw + t * GSZ, mK, Kt);
                    store_all<1>(gw + t * GSZ + NX * NU, mIK, IKt);
                } else {
                store_all(pw + t * PWSZ, mP, P);
                store_all(pw + t * PWSZ + NX * NX, mW, W2);
                RiccatiGains g;
                d4 P_old;
                ok = riccati_backward<M, false, false, true>(P, W2, Fc, Ys, Grep, Hxx, HX2, HU2, Hux0, scratch, lane, 1, g, P_old) && ok;
                pdall = pdall && g.pd;
                store_all<1>(gw + t * GSZ, mK, g.K);
                store_all<1>(gw + t * GSZ + NX * NU, mIK, g.IK);
                }
            };
            int tl = cnt - 1;
            for (; tl >= 1 && pdall; tl -= 2) { bstep(tl, Fa, Ya, Fb, Yb); if (pdall) bstep(tl - 1, Fb, Yb, Fa, Ya); else break; }
            if (tl == 0 && pdall) bstep(0, Fa, Ya, Fb, Yb);
            MS_T1(3);
        }
        pdall = pdall && ok;
        f_cur = wave_sum(a_f); th_cur = wave_sum(a_th); lamc = wave_sum(a_lc);
        inf_pr = wave_max(a_pr); inf_du = wave_max(a_du); zmax = wave_max(a_z); lmax = wave_max(a_l);
        finite = __all(fin) && (pdall ? (tile_finite(P) && tile_finite(W2)) : true);
        finite = __all(finite);
        MS_T1(1);
        return pdall;
    };

    // Forward pass of the LQ problem: dx, du, dlam into the workspace; returns grad(phi)' d = grad(L)' d + lambda' c  (A d = -c)
    auto forward = [&](double hs) -> double {
        MS_T0();
        wave_lds_sync();
        for (int i = lane; i < Mdl::SOLF_NCONST; i += 64) blk[1 + i] = Mdl::solf_const(i);
        Gather gFT, gGT, gE;
        make_gather(gFT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::solf_code(0, (c & 3) * NX + r) : -1)
                                                                               : ((r < NX && c < NX) ? Mdl::solf_code(0, c * NX + r) : -1); });
        make_gather(gGT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? Mdl::solf_code(1, (c & 3) * NU + r) : -1)
                                                                               : ((r < M && c < NX) ? Mdl::solf_code(1, c * NU + r) : -1); });
        make_gather(gE, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX && c == M) ? L::FC0 + r : -1; });
        d4 X2 = z;
        // dx, du, dlam leave through one buffer over the trajectory's [dx | du | dlam] block of the workspace
        const ColStore csX = make_col_store(NX, M, lane), csU = make_col_store(NU, M, lane);
        const auto rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dxb, 0, (int)(((int64_t)(T + 1) * NX + (int64_t)T * NU + (int64_t)T * NX) * 8), 0x00020000);
        const unsigned offU = (unsigned)((T + 1) * NX) * 8u, offL = offU + (unsigned)(T * NU) * 8u;
        // gains and P_{t+1}, W_{t+1} of step t are requested one step ahead (each lane re-reads what it stored in the backward sweep)
        d4 KTn = -load_all<NRT>(gw, mKT);
        d4 kn = -load_all<1>(gw + NX * NU, mIK);
        d4 Pq = load_all<NRT>(pw, mPld), Wq = load_all<NRT>(pw + NX * NX, mW);
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            wave_lds_sync();
            MS_T1(4);
            if (lane < cnt) {
                PDP_MS_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU];
                double* row = pool + lane * L::FSTRIDE;
#pragma unroll
                for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; row[L::FC0 + i] = hs * cst[t * NX + i]; }
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                PackedSink s{row};
                Mdl::eval_solf(xc, uc, nullptr, th, pc, s);
            }
            wave_lds_sync();
            MS_T1(5);
            GatherRun rFT = gather_at(gFT, 0, blk), rGT = gather_at(gGT, 0, blk), rE = gather_at(gE, 0, blk);
            auto fstep = [&](int tl, const d4 Xc, d4& Xn, const d4 KTc, const d4 kc, const d4 Pc, const d4 Wc, d4& KTnx, d4& knx, d4& Pnx, d4& Wnx) {
                const int t = t0 + tl, tnx = (t + 1 < T) ? t + 1 : t;
                KTnx = -load_all<NRT>(gw + tnx * GSZ, mKT);
                knx = -load_all<1>(gw + tnx * GSZ + NX * NU, mIK);
                Pnx = load_all<NRT>(pw + tnx * PWSZ, mPld);
                Wnx = load_all<NRT>(pw + tnx * PWSZ + NX * NX, mW);
                d4 FT = gather_run<NRT>(rFT, 1);
                d4 GT = gather_run<1>(rGT, 1);
                d4 E2 = gather_run<NRT>(rE, 1);
                d4 U2, Lm;
                if constexpr (SMALL) {
                    U2 = z; Xn = z; Lm = z;
                    double u0, x1;
                    riccati_small_forward(KTc[0], kc[0], FT[0], GT[0], E2[0], Xc[0], u0, x1);
                    U2[0] = u0; Xn[0] = x1;
                    Lm[0] = mma4_blk(Pc[0], x1, Wc[0]);
                } else {
                riccati_forward(KTc, kc, FT, GT, E2, Xc, U2, Xn);
                Lm = mma_tn(Pc, Xn, Wc);                             // dlam_t = P_{t+1} dx_{t+1} + W_{t+1}   (P symmetric)
                }
                store_tile_column_buf<1>(rsD, offU + (unsigned)(t * NU) * 8u, csU, U2);
                store_tile_column_buf<NRT>(rsD, (unsigned)((t + 1) * NX) * 8u, csX, Xn);
                store_tile_column_buf<NRT>(rsD, offL + (unsigned)(t * NX) * 8u, csX, Lm);
            };
            d4 Xb, KTb, kb, Pb, Wb;
            int tl = 0;
            for (; tl + 1 < cnt; tl += 2) { fstep(tl, X2, Xb, KTn, kn, Pq, Wq, KTb, kb, Pb, Wb); fstep(tl + 1, Xb, X2, KTb, kb, Pb, Wb, KTn, kn, Pq, Wq); }
            if (tl < cnt) { fstep(tl, X2, Xb, KTn, kn, Pq, Wq, KTb, kb, Pb, Wb); X2 = Xb; KTn = KTb; kn = kb; Pq = Pb; Wq = Wb; }
            MS_T1(6);
        }
        __threadfence_block();
        wave_lds_sync();
        double gd = 0.0;
        for (int q = lane; q < T * NX; q += 64) gd += rxs[NX + q] * dxb[NX + q];
        for (int q = lane; q < T * NU; q += 64) gd += rus[q] * dub[q];
        gd = wave_sum(gd) + lamc;
        MS_T1(4);
        return gd;
    };

    // objective and constraint violation of the trial point (x + a dx, u + a du): lane = stage
    auto trial = [&](double a, double& ft, double& tht) {
        PDP_MS_PAR();
        double sf = 0.0, st = 0.0;
        for (int t = lane; t < T; t += 64) {
            double xc[NX], uc[NU], xn[NX], v[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i] + a * dxb[t * NX + i]; xn[i] = xb[(t + 1) * NX + i] + a * dxb[(t + 1) * NX + i]; }
#pragma unroll
            for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i] + a * dub[t * NU + i];
            Mdl::dyn(xc, uc, th, pc, v);
#pragma unroll
            for (int i = 0; i < NX; ++i) st += fabs(v[i] - xn[i]);
            sf += Mdl::path_cost(xc, uc, th, pc);
            if (t == T - 1) sf += Mdl::final_cost(xn, th, pc);
        }
        ft = wave_sum(sf);
        tht = wave_sum(st);
    };

    // KKT residuals of the current point with one lane per stage (no matrices, no sweep): what the convergence test needs.  The
    // iteration tests convergence BEFORE it pays for a sweep (IPOPT's order), so a solve costs one sweep per Newton step, not one more.
    auto residuals = [&]() {
        PDP_MS_PAR();
        double a_f = 0.0, a_th = 0.0, a_pr = 0.0, a_du = 0.0, a_z = 0.0, a_l = 0.0;
        bool fin = true;
        for (int t = lane; t < T; t += 64) {
            double xc[NX], uc[NU], lc[NX], v[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; lc[i] = lb[t * NX + i]; a_z = fmax(a_z, fabs(xc[i])); a_l = fmax(a_l, fabs(lc[i])); }
#pragma unroll
            for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; a_z = fmax(a_z, fabs(uc[i])); }
            Mdl::dyn(xc, uc, th, pc, v);
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double ci = v[i] - xb[(t + 1) * NX + i]; a_th += fabs(ci); a_pr = fmax(a_pr, fabs(ci)); fin = fin && fabs(ci) <= 1.7e308; }
            Mdl::costate_step(xc, uc, lc, th, pc, v);
#pragma unroll
            for (int i = 0; i < NX; ++i) { const double g = t > 0 ? v[i] - lb[(t - 1) * NX + i] : 0.0; a_du = fmax(a_du, fabs(g)); fin = fin && fabs(g) <= 1.7e308; }
            double hu[NU];
            Mdl::dHu(xc, uc, lc, th, pc, hu);
#pragma unroll
            for (int i = 0; i < NU; ++i) { a_du = fmax(a_du, fabs(hu[i])); fin = fin && fabs(hu[i]) <= 1.7e308; }
            a_f += Mdl::path_cost(xc, uc, th, pc);
            if (t == T - 1) {
                double xT[NX], hx[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { xT[i] = xb[T * NX + i]; a_z = fmax(a_z, fabs(xT[i])); }
                Mdl::dhx(xT, th, pc, hx);
                a_f += Mdl::final_cost(xT, th, pc);
#pragma unroll
                for (int i = 0; i < NX; ++i) { const double g = hx[i] - lc[i]; a_du = fmax(a_du, fabs(g)); fin = fin && fabs(g) <= 1.7e308; }
            }
        }
        f_cur = wave_sum(a_f); th_cur = wave_sum(a_th);
        inf_pr = wave_max(a_pr); inf_du = wave_max(a_du); zmax = wave_max(a_z); lmax = wave_max(a_l);
        finite = __all(fin);
    };

    // ---- main loop.  Every trip runs ONE backward sweep (the lambdas above have a single call site each: one copy of the sweep code).
    // phase 0 (cold start only): IPOPT's least-squares multiplier estimate (constr_mult_init_max = 1000),
    //     [I A'; A 0] [w; lambda] = -[grad f; 0]  - the same sweep with W = I and no defects; phase 1: the iteration.
#ifdef PDP_MS_TIMING
    tmi = __builtin_readcyclecounter();
#endif
    int st = 0, it = 0, nfilt = 0, conv = 0, phase = warm ? 1 : 0;
    double hs = warm ? 1.0 : 0.0, dw = warm ? 0.0 : 1.0, dw_last = 0.0, theta_max = 0.0, theta_min = 0.0;
    for (;;) {
        if (phase == 1 && dw == 0.0) {              // a new iterate: converged?  (a sweep follows only if not - or once more for the gains output)
            MS_T0();
            residuals();
            MS_T1(0);
            if (!finite) { st |= PDP_STATUS_NONFINITE; break; }
            if (it == 0) { theta_max = 1e4 * fmax(1.0, th_cur); theta_min = 1e-4 * fmax(1.0, th_cur); }
            if (inf_pr <= op.tol * (1.0 + zmax) && inf_du <= op.tol * (1.0 + lmax)) { conv = 1; if (!gains_out) break; }
            else if (it >= op.max_iter) { st |= PDP_MS_MAXITER; break; }
        }
        const bool pd = backward(hs, dw);
        if (conv) break;                            // (the sweep at the solution left the LQR gains in the workspace)
        if (phase == 0) {
            if (!(pd && finite)) { phase = 1; hs = 1.0; dw = 0.0; continue; }
        } else {
            if (!finite) { st |= PDP_STATUS_NONFINITE; break; }
            if (!pd) {                              // Algorithm IC (defaults: 1e-4 first, x100 / x8 up, /3 down, 1e20 max)
                if (dw == 0.0) dw = dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last * (1.0 / 3.0));
                else dw *= dw_last == 0.0 ? 100.0 : 8.0;
                if (dw > 1e20) { st |= PDP_MS_INERTIA; break; }
                continue;
            }
            if (dw > 0.0) dw_last = dw;
        }
        const double gd = forward(hs);
        if (phase == 0) {
            double lm = 0.0;
            bool fin = true;
            for (int q = lane; q < T * NX; q += 64) { const double v = dlb[q]; lm = fmax(lm, fabs(v)); fin = fin && fabs(v) <= 1.7e308; }
            lm = wave_max(lm);
            if (__all(fin) && lm <= 1000.0) {
                for (int q = lane; q < T * NX; q += 64) lb[q] = dlb[q];
                __threadfence_block();
                wave_lds_sync();
            }
            phase = 1; hs = 1.0; dw = 0.0;
            continue;
        }
        const double f = f_cur, theta = th_cur;
        // backtracking filter line search (Algorithm A): alpha_min below which IPOPT would enter the restoration phase
        double amin = 1e-5;
        if (gd < 0.0) {
            amin = fmin(1e-5, 1e-8 * theta / (-gd));
            if (theta <= theta_min) amin = fmin(amin, pow(theta, 1.1) / pow(-gd, 2.3));
        }
        amin *= 0.05;
        double alpha = 1.0, ft = 0.0, tht = 0.0;
        bool accepted = false, ftype = false;
        MS_T0();
        while (alpha >= amin) {
            trial(alpha, ft, tht);
            bool okf = fabs(ft) <= 1.7e308 && fabs(tht) <= 1.7e308 && tht <= theta_max;
            if (okf) {
                bool dominated = false;
                for (int e = lane; e < nfilt; e += 64) dominated = dominated || (tht >= fth[e] && ft >= fph[e]);
                okf = !__any(dominated);
            }
            if (okf) {
                const bool switching = gd < 0.0 && alpha * pow(-gd, 2.3) > pow(theta, 1.1);
                if (theta <= theta_min && switching) {
                    if (ft <= f + 1e-8 * alpha * gd + 10.0 * 2.220446049250313e-16 * fabs(f)) { accepted = true; ftype = true; }
                } else if (tht <= (1.0 - 1e-5) * theta || ft <= f - 1e-8 * theta) accepted = true;
            }
            if (accepted) break;
            alpha *= 0.5;
        }
        MS_T1(7);
#ifdef PDP_MS_TIMING
        if (iter_log && it < op.log_rows && lane == 0) {
            double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
            const long long now_ = __builtin_readcyclecounter();
            // residuals | backward: terminal + reductions | backward evaluation | Riccati steps | forward evaluation + gd | (unused) | forward steps | line search; total of the iteration
            row[0] = (double)tm[0]; row[1] = (double)tm[1]; row[2] = (double)tm[2]; row[3] = (double)tm[3]; row[4] = (double)(tm[4] + tm[5]); row[5] = (double)tm[6]; row[6] = (double)tm[7];
            row[7] = (double)(now_ - tmi);
            tmi = now_;
            for (int k_ = 0; k_ < 8; ++k_) tm[k_] = 0;
        }
#else
        if (iter_log && it < op.log_rows && lane == 0) {
            double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
            row[0] = it; row[1] = f; row[2] = inf_pr; row[3] = inf_du; row[4] = dw; row[5] = accepted ? alpha : 0.0; row[6] = gd; row[7] = theta;
        }
#endif
        if (!accepted) { st |= PDP_MS_RESTORATION; break; }
        if (!ftype) {                               // (at most one entry per iteration: the workspace holds max_iter + 1)
            if (lane == 0) { fth[nfilt] = (1.0 - 1e-5) * theta; fph[nfilt] = f - 1e-8 * theta; }
            ++nfilt;
        }
        for (int q = lane; q < T * NX; q += 64) { xb[NX + q] += alpha * dxb[NX + q]; lb[q] += alpha * dlb[q]; }
        for (int q = lane; q < T * NU; q += 64) ub[q] += alpha * dub[q];
        __threadfence_block();
        wave_lds_sync();
        dw = 0.0;
        ++it;
    }
    if (lane == 0) {
        if (cost) cost[b] = f_cur;
        if (resid) { resid[2 * b] = inf_pr; resid[2 * b + 1] = inf_du; }
        if (converged) converged[b] = conv;
        if (iters) iters[b] = it;
        if (status) status[b] = st;
    }
    if (gains_out) {        // LQR feedback around the last linearisation point, in the layout of pdp_oc_rollout_feedback_batched: K^T [n][m] | k [m]
        __threadfence_block();
        wave_lds_sync();
        constexpr int G2 = NX * NU + NU;
        double* go = gains_out + (int64_t)b * T * G2;
        for (int q = lane; q < T * G2; q += 64) {
            const int t = q / G2, r = q - t * G2;
            go[q] = r < NX * NU ? gw[t * GSZ + (r % NU) * NX + r / NU] : gw[t * GSZ + r];
        }
    }
#undef PDP_MS_PAR
#undef MS_T0
#undef MS_T1
}

template <class Mdl>
__host__ inline size_t ms_lds_bytes() { return sizeof(double) * (size_t)MsLayout<Mdl>::LDS_DOUBLES; }

}  // namespace pdp
