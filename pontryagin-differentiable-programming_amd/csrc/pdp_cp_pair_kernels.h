// pdp_cp_pair_kernels.h - fused ControlPlanning.step for the Lagrange-polynomial policy (reference PDP/PDP.py:686-725 setPolyPolicy, 763-786 integrateSys,
// 826-834 integrateAuxSys, 850-878 step) as a PIPELINE of two wavefronts per trajectory: cp_step_poly2_kernel.  Same arithmetic in the same order as
// cp_step_poly_kernel (pdp_model_kernels.h) - loss, trajectory and gradient bit-identical (tests/test_gpu_cp_pair.py).
//
// Why.  Both halves of this step run FORWARD in time: the rollout x_{t+1} = f(x_t, pi(t, theta)) and the sensitivity recursion
// X_{t+1} = F_t X_t + G_t dpi/dtheta.  The one-wave kernel ran them one after the other - a scalar recursion (one dependent fp64 operation every 7 - 11
// cycles, the MFMA pipe idle) and then chunks of Jacobian evaluation + MFMA chains (the scalar pipe mostly idle).  Here
//     wave R  rolls the trajectory out into the LDS staging and publishes how far it has got (one LDS counter, a plain store per step),
//             then writes the trajectory, the loss and the terminal gradient;
//     wave S  follows one chunk behind: Jacobians of the chunk (lane = stage) into the packed pool, then the sensitivity recursion over the chunk on the MFMA tiles.
// The launch lasts max(rollout, sensitivities) + one chunk of rollout instead of their sum, and the two instruction streams - one latency-bound scalar, one
// MFMA - are the kind that interleave well on one SIMD.  TPW trajectories per workgroup as in pdp_fused3_kernels.h: 4 (waves w and w + 4
// share a SIMD) once the batch fills the chip, 1 or 2 (the pair on two SIMDs) below.
#pragma once
#include "pdp_fused3_kernels.h"

namespace pdp {

// doubles of LDS per trajectory (layout of cp_step_poly_kernel + the hand-over counter)
template <class Mdl>
__host__ __device__ inline int cp_pair_slice(int T, int n_pivots) {
    const int n = 1 + Mdl::PATH_NCONST + Mdl::CHUNK * (Mdl::PATH_NVAR | 1) + (T + 1) * Mdl::NX + T * Mdl::NU + T * n_pivots + Mdl::NX + 8 + 64 +
                  (Mdl::NX > Mdl::NU ? Mdl::NX : Mdl::NU) + 8;
    return (n + 1) & ~1;
}

template <class Mdl, int NT, int TPW>
__global__ void __launch_bounds__(128 * TPW) cp_step_poly2_kernel(int B, int T, pdp_policy pol, int p, const double* __restrict__ x0,
                                                                   const double* __restrict__ theta, int tb, double* __restrict__ loss,
                                                                   double* __restrict__ grad, double* __restrict__ xo, double* __restrict__ uo, int slice) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, CH = Mdl::CHUNK, M = NU;
    constexpr int NC = 1 + Mdl::PATH_NCONST, STRIDE = Mdl::PATH_NVAR | 1;
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "trajectories per workgroup");
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = wid & (TPW - 1);
    const bool roller = wid < TPW;
    const int b = blockIdx.x * TPW + slot;
    double* blk = lds_all + (size_t)slot * slice;      // [cpool | pool]
    double* pool = blk + NC;
    double* xs = pool + CH * STRIDE;         // (T+1) x NX
    double* us = xs + (T + 1) * NX;          // T x NU
    const int np = pol.n_pivots;
    double* basis = us + T * NU;             // T x n_pivots
    double* hx = basis + T * np;             // NX
    double* dump = hx + NX + 8;              // 64 + max(NX, NU) words nobody reads (see the rollout)
    int* fl = (int*)(dump + 64 + (NX > NU ? NX : NU));       // [0]: stages rolled out so far (T + 1: the terminal gradient is in hx as well)
    if (roller && lane == 0) fl[0] = 0;
    const int bb = b < B ? b : B - 1;                         // (a workgroup's spare slots repeat the last trajectory and store nothing)
    const bool mine = b < B;
    if (roller) {
        for (int t = lane; t < T; t += 64)
            for (int i = 0; i < np; ++i) basis[t * np + i] = lagrange_basis(pol, i, (double)t);
        if (lane == 0) blk[0] = 0.0;
        for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
    }
    __syncthreads();                                          // basis, constants and the zeroed counter (two workgroup barriers in all, both before the pipeline starts)
    const double* th = theta + (int64_t)bb * tb;
    // the controls: the policy is open-loop, u_t = sum_i b_i(t) theta_i - all of them at once, lane = time step, by wave R (see cp_step_poly_kernel)
    if (roller) {
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
    }
    __syncthreads();
    double pc[Mdl::NPC];
    Mdl::precompute(nullptr, pc);
    const int nchunk = (T + CH - 1) / CH;
    const int ch = (T + nchunk - 1) / nchunk;                // chunks of equal length
    if (roller) {
        // ================================================ wave R: rollout ================================================
        double J = 0.0;
        double xc[NX], xn[NX], uc[NU], un[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)bb * NX + i];
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
            // x_{t+1} into the LDS staging from lane 0 without a conditional block (see cp_step_poly_kernel)
            {
                double* dx_ = lane == 0 ? xs + (t + 1) * NX : dump + lane;
#pragma unroll
                for (int i = 0; i < NX; ++i) dx_[i] = xn[i];
            }
            // progress for wave S, every step and WITHOUT a branch or a wait (a conditional block here would make the waits of every later step conservative): the
            // LDS unit executes one wave's operations in issue order, so whoever reads this counter afterwards finds the staging stores above done
            asm volatile("" ::: "memory");
            __hip_atomic_store(fl, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        J += Mdl::final_cost(xc, nullptr, pc);
        double h[NX];
        Mdl::dhx(xc, nullptr, pc, h);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) hx[i] = h[i];
        }
        f3_signal(fl, T + 1);
        if (mine && blockIdx.y == 0) {
            if (xo) for (int i = lane; i < (T + 1) * NX; i += 64) xo[(int64_t)b * (T + 1) * NX + i] = xs[i];
            if (uo) for (int i = lane; i < T * NU; i += 64) uo[(int64_t)b * T * NU + i] = us[i];
            if (lane == 0) loss[b] = J;
        }
    } else {
        // ================================================ wave S: forward sensitivities ================================================
        const d4 z = zero4();
        const int tile0 = blockIdx.y * NT;
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
        int piv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { const int cidx = 16 * (tile0 + j) + col; piv[j] = (cidx < p && row0 < M && (cidx % NU) == row0) ? cidx / NU : -1; }
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            f3_wait_ge(fl, t0 + cnt);                        // acquire: x_t, u_t of the chunk are in the staging
            wave_lds_sync();
            if (lane < cnt) {
                const int t = t0 + lane;
                double xc[NX], uc[NU];
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xs[t * NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = us[t * NU + i];
                PackedSink s{pool + lane * STRIDE};
                Mdl::eval_path(xc, uc, nullptr, nullptr, pc, s);
            }
            wave_lds_sync();
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
        f3_wait_ge(fl, T + 1);                               // the terminal gradient h_x(x_T)
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc[j] += hx[row] * X[j][r]; }
            double a = sum_over_rowgroups(acc[j]);
            if (mine && lane < 16 && 16 * (tile0 + j) + lane < p) grad[(int64_t)b * p + 16 * (tile0 + j) + lane] = a;
        }
    }
}

// Fused SysID.step (reference PDP/PDP.py:1178-1296: integrateSys with the given controls, getAuxSys, integrateAuxSys X_{t+1} = F X_t + E, chain rule with the
// prediction error) as the same two-wave pipeline: wave R rolls the model out along the recorded controls, wave S follows a chunk behind with the Jacobians, the
// prediction errors and the sensitivity recursion.  Same arithmetic in the same order as sysid_step_kernel (bit-identical, tests/test_gpu_cp_pair.py).
template <class Mdl, int NT, int TPW>
__global__ void __launch_bounds__(128 * TPW) sysid_step2_kernel(int B, int T, const double* __restrict__ u, const double* __restrict__ xobs,
                                                                 const double* __restrict__ theta, int tb, double* __restrict__ loss, double* __restrict__ grad, int slice) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = Mdl::CHUNK;
    constexpr int NC = 1 + Mdl::PATH_NCONST, DLX = Mdl::PATH_NVAR, STRIDE = (Mdl::PATH_NVAR + NX) | 1;
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "trajectories per workgroup");
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = wid & (TPW - 1);
    const bool roller = wid < TPW;
    const int b = blockIdx.x * TPW + slot;
    const int bb = b < B ? b : B - 1;
    const bool mine = b < B;
    double* blk = lds_all + (size_t)slot * slice;
    double* pool = blk + NC;
    double* xs = pool + CH * STRIDE;         // (T+1) x NX
    double* dlT = xs + (T + 1) * NX;         // NX
    double* us = dlT + NX + 1;               // T x NU
    double* dump = us + T * NU;              // 64 + NX
    int* fl = (int*)(dump + 64 + NX);        // stages rolled out so far
    const double* ub = u + (int64_t)bb * T * NU;
    const double* ob = xobs + (int64_t)bb * (T + 1) * NX;
    if (roller) {
        if (lane == 0) { blk[0] = 0.0; fl[0] = 0; }
        for (int i_ = lane; i_ < Mdl::PATH_NCONST; i_ += 64) blk[1 + i_] = Mdl::path_const(i_);
        for (int q = lane; q < T * NU; q += 64) us[q] = ub[q];
    }
    __syncthreads();                                          // the only workgroup barrier: constants, controls and the zeroed counter
    double th[NP];
    load_theta<Mdl>(theta, bb, tb, th);
    double pc[Mdl::NPC];
    Mdl::precompute(th, pc);
    if (roller) {
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
            asm volatile("" ::: "memory");                   // progress for wave S: a plain LDS store behind the staging stores (see cp_step_poly2_kernel)
            __hip_atomic_store(fl, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        const d4 z = zero4();
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
        const int ch = (T + nchunk - 1) / nchunk;
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            f3_wait_ge(fl, t0 + cnt);                        // x_t of the chunk's stages are in the staging (x_{t0+cnt-1} was stored by step t0+cnt-2; the counter is past it)
            wave_lds_sync();
            if (lane < cnt) {
                const int t = t0 + lane;
                double xc[NX], uc[NU];
                double* row = pool + lane * STRIDE;
#pragma unroll
                for (int i = 0; i < NX; ++i) { xc[i] = xs[t * NX + i]; double d = xc[i] - ob[t * NX + i]; row[DLX + i] = d; lsum += d * d; }
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = us[t * NU + i];
                PackedSink s{row};
                Mdl::eval_path(xc, uc, nullptr, th, pc, s);
            }
            wave_lds_sync();
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
        f3_wait_ge(fl, T);                                   // x_T
        wave_lds_sync();
        if (lane < NX) { double d = xs[T * NX + lane] - ob[T * NX + lane]; dlT[lane] = d; lsum += d * d; }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc[j] += dlT[row] * X[j][r]; }
            double a = sum_over_rowgroups(acc[j]);
            if (mine && lane < 16 && 16 * j + lane < NP) grad[(int64_t)b * NP + 16 * j + lane] = a;
        }
        lsum = wave_sum(lsum);
        if (mine && lane == 0) loss[b] = lsum;
    }
}

}  // namespace pdp
