// pdp_fused2_kernels.h - the fused OC unit (forward + costates + aux system in LDS + Riccati + PDP gradient) with TWO wavefronts per
// trajectory: oc_pdp_fused2_kernel.  Same inputs, outputs, workspace and arithmetic as oc_pdp_fused_kernel (pdp_model_kernels.h).
//
// Why.  At the headline batch (1024 trajectories = the 1024 SIMDs of an MI355X) the one-wave kernel is latency-bound: a trajectory is ONE
// serial instruction stream of ~280 k cycles of which 98 k are MFMA, and nothing else runs on its SIMD.  Measured this round
// (profiles/r02_probe_two_waves_per_simd.txt): two waves on a SIMD do overlap MFMA with VALU / LDS / waits - but a second TRAJECTORY per
// SIMD does not exist at B = 1024.  So the trajectory itself is cut in two: the backward Riccati step has two independent halves,
//     wave A:  PF = P F            Qux = Hux + G' PF        Pn = Hxx + F' PF          | K = Z Qux      P- = Pn - Qux' K
//     wave B:  PY2 = P [G|E] + W   FY = [Hxu|Hxe] + F' PY2  [Quu|Que] = HU2 + G' PY2  | Z = Quu^-T     [I|k] = Z [Quu|Que]   W- = Wn - Qux' k
// that meet twice per step through 64-double LDS mailboxes (Z and Qux; then P-, which goes through LDS anyway for the symmetrisation
// P <- (P + P')/2 and is read back by BOTH waves).  9 full + 5 small MFMAs per wave and step instead of 18 + 10 in one stream; half the tile
// registers per wave, so two workgroups' waves share a SIMD (256 registers per wave) and fill each other's gaps.  The serial phases
// (rollout, lane-per-step evaluation, costate chain, forward sweep) stay on wave A; wave B evaluates the Hessian group.
#pragma once
#include "pdp_model_kernels.h"

namespace pdp {

// workgroup barrier for two waves that exchange through LDS: DS results waited for, global stores NOT (unlike __syncthreads())
PDP_DEV void wg_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Z = Quu^-T in rep form (the m x m block replicated in the four column blocks): the m x m solve of riccati_backward (pdp_riccati.h), same code
template <int M>
PDP_DEV double quu_inverse_rep(double Q20, double* scratch, int lane, bool& ok) {
    const int row = lane >> 4, col = lane & 15;
    double Zrep = 0.0;
    if constexpr (M == 4) {
        scratch[544 + lane] = Q20;
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
        const double ac = Q20 * cof;
        const double t0 = readlane_f64(ac, 0), t1 = readlane_f64(ac, 1), t2 = readlane_f64(ac, 2), t3 = readlane_f64(ac, 3);
        const double det = (t0 + t1) + (t2 + t3), mag = (fabs(t0) + fabs(t1)) + (fabs(t2) + fabs(t3));
        if (fabs(det) > 1e-10 * mag && fabs(det) <= 1.7e308) {
            double rdet = __builtin_amdgcn_rcp(det);
            rdet = fma(fma(-det, rdet, 1.0), rdet, rdet);
            Zrep = cof * rdet;
        } else {
            double a[16], ai[16];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a[ii * 4 + jj] = readlane_f64(Q20, 16 * ii + jj);
            ok = inverse_small<4>(a, ai) && ok;
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
            for (int j = 0; j < M; ++j) a[i * M + j] = readlane_f64(Q20, 16 * i + j);
        ok = inverse_small_fast<M>(a, ai) && ok;
        double zz = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = 0; j < M; ++j) zz = (row == i && (col & 3) == j) ? ai[j * M + i] : zz;
        Zrep = zz;
    }
    return Zrep;
}

PDP_DEV d4 tile_from_lds17(const double* s, int lane) {
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = s[tile_row(lane, r) * 17 + tile_col(lane)];
    return v;
}

template <class Mdl>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
oc_pdp_fused2_kernel(int B, int T, int flags, const double* __restrict__ x0, const double* __restrict__ u, const double* __restrict__ theta, int tb,
                     const double* __restrict__ demo_x, const double* __restrict__ demo_u, double* __restrict__ x, double* __restrict__ lam,
                     double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ dxdp, double* __restrict__ dudp,
                     int32_t* __restrict__ status, double* __restrict__ ws_gain) {
    using L = FusedLayout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, CH = L::CH, M = NU;
    constexpr int GSZ = fused_gain_doubles<Mdl>();
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* scratch = lds;                              // [0,272) P- (stride 17) | [272,336) Z | [336,400) Qux | [400,408) flags | [544,608) Quu rows
    double* Pbuf = scratch;
    double* Zx = scratch + 272;
    double* Qx = scratch + 336;
    double* fl = scratch + 400;
    double* blk = lds + RICCATI_SCRATCH;                // [constants (NC) | pool]
    double* pool = blk + L::NC;
    double* dlT = pool + fused_pool_doubles<Mdl>(T);
    double* par = dlT + NX;
    const int b = blockIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool waveA = wid == 0;
    const d4 z = zero4();
    if (waveA) {
        double th0[NP], pc0[Mdl::NPC];
        load_theta<Mdl>(theta, b, tb, th0);
        Mdl::precompute(th0, pc0);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NP; ++i) par[i] = th0[i];
#pragma unroll
            for (int i = 0; i < Mdl::NPC; ++i) par[NP + i] = pc0[i];
            fl[0] = 1.0; fl[1] = 1.0;                   // wave B's ok / finite flags
        }
    }
    wg_sync();
#define PDP_F2_PAR()                                                \
    double th[NP], pc[Mdl::NPC];                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) th[i_] = par[i_]; \
    _Pragma("unroll") for (int i_ = 0; i_ < Mdl::NPC; ++i_) pc[i_] = par[NP + i_]
    double* xb = x + (int64_t)b * (T + 1) * NX;
    double* lb = lam + (int64_t)b * T * NX;
    const double* ub = u + (int64_t)b * T * NU;
    double* gw = ws_gain + (int64_t)b * T * GSZ;
    const bool given = (flags & PDP_OC_GIVEN_TRAJ) != 0;
#ifdef PDP_PHASE_TIMING     // debug builds (probes/phase_timing2.py): cycle stamps of workgroup 0 behind loss[B]
    long long ts[8], fs[8], acc_t[4] = {0, 0, 0, 0};
    int nts = 0;
    for (int i = 0; i < 8; ++i) { ts[i] = 0; fs[i] = 0; }
#define F2_STAMP() ts[nts++] = __builtin_readcyclecounter()
#define F2_FINE(i, cond) if (cond) fs[i] = __builtin_readcyclecounter()
#define F2_ACC(k, t_) acc_t[k] += __builtin_readcyclecounter() - (t_)
#define F2_NOW() __builtin_readcyclecounter()
#else
#define F2_STAMP()
#define F2_FINE(i, cond)
#define F2_ACC(k, t_)
#define F2_NOW() 0
#endif
    F2_STAMP();

    // ---------------- rollout (wave A), staged in the still unused pool, written out coalesced --------------------------------
    if (!given && waveA) {
        double* xs = pool;
        double* us = pool + (T + 1) * NX;
        for (int i = lane; i < T * NU; i += 64) us[i] = ub[i];
        PDP_F2_PAR();
        double xc[NX], xn[NX], uc[NU];
#pragma unroll
        for (int i = 0; i < NX; ++i) xc[i] = x0[(int64_t)b * NX + i];
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = xc[i];
        }
        wave_lds_sync();
        double un[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) un[i] = us[i];
        for (int t = 0; t < T; ++t) {
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
            Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
            for (int i = 0; i < NX; ++i) xc[i] = xn[i];
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < NX; ++i) xs[(t + 1) * NX + i] = xn[i];
            }
        }
        wave_lds_sync();
        for (int i = lane; i < (T + 1) * NX; i += 64) xb[i] = xs[i];
        __threadfence_block();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // x is re-read below by both waves
    }
    wg_sync();
    F2_STAMP();

    // ---------------- terminal condition: P = hxx(x_T) (both waves), W = hxe(x_T) (wave B), lambda_T (wave A) -------------------
    bool ok = true;
    d4 P, W2 = z, Lam = z;
    {
        if (waveA) {
            if (lane == 0) blk[0] = 0.0;
            for (int i_ = lane; i_ < Mdl::FIN_NCONST; i_ += 64) blk[1 + i_] = Mdl::fin_const(i_);
            if (lane == 0) {
                PDP_F2_PAR();
                double xT[NX], lT[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) xT[i] = xb[T * NX + i];
                PackedSink s{pool};
                Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
                Mdl::dhx(xT, th, pc, lT);
#pragma unroll
                for (int i = 0; i < NX; ++i) dlT[i] = lT[i];
            }
        }
        wg_sync();
        Gather gP, gW;
        make_gather(gP, lane, L::NC, 0, [](int r, int c) { return (r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1; });
        make_gather(gW, lane, L::NC, 0, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fin_code(1, r * NP + (c - M)) : -1; });
        P = gather_tile(blk, gP, 0);
        if (!waveA) W2 = gather_tile(blk, gW, 0);
        if (waveA && !given) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int row = tile_row(lane, r); if (row < NX && tile_col(lane) == 0) Lam[r] = dlT[row]; }
        }
        wg_sync();
    }

    F2_STAMP();
    // ---------------- backward sweep ------------------------------------------------------------------------------------------------
    {
        if (waveA) {
            for (int i_ = lane; i_ < Mdl::PATHA_NCONST; i_ += 64) blk[1 + i_] = Mdl::patha_const(i_);
            for (int i_ = lane; i_ < Mdl::PATHB_NCONST; i_ += 64) blk[1 + Mdl::PATHA_NCONST + i_] = Mdl::pathb_const(i_);
        }
        constexpr int NA = L::NA, NCA = Mdl::PATHA_NCONST;
        auto codeA = [](int mat, int i) { return Mdl::patha_code(mat, i); };
        auto codeB = [](int mat, int i) { int c = Mdl::pathb_code(mat, i); return c >= 0 ? c + NA : (c == -1 ? -1 : c - NCA); };
        Gather gF, gGr, g3, g4, g5;                     // wave A: F, Grep, Hux (1 reg), Hxx, c_x ;  wave B: F, Grep, [G|E], [Hxu|Hxe], [Huu|Hue] (1 reg)
        make_gather(gF, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && c < NX) ? codeA(0, r * NX + c) : -1; });
        make_gather(gGr, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeA(1, r * NU + (c & 3)) : -1; });
        if (waveA) {
            make_gather(g3, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < M && c < NX) ? codeB(1, c * NU + r) : -1; });             // Hux
            make_gather(g4, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && c < NX) ? codeB(0, r * NX + c) : -1; });            // Hxx
            make_gather(g5, lane, L::NC, L::BSTRIDE, [&](int r, int c) { return (r < NX && c == 0) ? codeA(3, r) : -1; });                     // c_x
        } else {
            make_gather(g3, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
                return r >= NX ? -1 : (c < M ? codeA(1, r * NU + c) : (c < M + NP ? codeA(2, r * NP + (c - M)) : -1)); });                     // [G|E]
            make_gather(g4, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
                return r >= NX ? -1 : (c < M ? codeB(1, r * NU + c) : (c < M + NP ? codeB(2, r * NP + (c - M)) : -1)); });                     // [Hxu|Hxe]
            make_gather(g5, lane, L::NC, L::BSTRIDE, [&](int r, int c) {
                return r >= M ? -1 : (c < M ? codeB(3, r * NU + c) : (c < M + NP ? codeB(4, r * NP + (c - M)) : -1)); });                      // [Huu|Hue]
        }
        const TileMapBytes mK = make_tile_map_sink(NU, NX, NX, 0, 0, lane, fused_gain0_doubles<Mdl>() - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;
        for (int c = nchunk - 1; c >= 0; --c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            wg_sync();                                  // the pool of the previous chunk is no longer read
            [[maybe_unused]] const long long tc0 = F2_NOW();
            if (waveA && lane < cnt) {                  // (A) lane = time step: F, G, E, c_x at (x_t, u_t)
                PDP_F2_PAR();
                const int t = t0 + lane;
                double xc[NX], uc[NU];
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xb[t * NX + i];
#pragma unroll
                for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                PackedSink s{pool + lane * L::BSTRIDE};
                Mdl::eval_patha(xc, uc, nullptr, th, pc, s);
            }
            if (waveA && !given) {                      // (C) costates through the chunk on MFMA; pool row tl receives lambda_{t+1}
                wave_lds_sync();
                GatherRun cF = gather_at(gF, cnt - 1, blk), cC = gather_at(g5, cnt - 1, blk);
                for (int tl = cnt - 1; tl >= 0; --tl) {
                    const d4 Fc = gather_run(cF, -1), CX = gather_run(cC, -1);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int row = tile_row(lane, r); if (row < NX && tile_col(lane) == 0) pool[tl * L::BSTRIDE + L::LAM + row] = Lam[r]; }
                    Lam = mma_tn(Fc, Lam, CX);          // lambda_t = c_x(x_t,u_t) + F_t' lambda_{t+1}
                }
            }
            wg_sync();
            F2_ACC(0, tc0);
            [[maybe_unused]] const long long tc1 = F2_NOW();
            if (!waveA && lane < cnt) {                 // (B) lane = time step (wave B): Hamiltonian Hessians at (x_t, u_t, lambda_{t+1})
                PDP_F2_PAR();
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
            wg_sync();
            F2_ACC(1, tc1);
            [[maybe_unused]] const long long tc2 = F2_NOW();
            // Riccati steps: two halves per step, meeting at the Z / Qux mailboxes and at the symmetrisation buffer
            GatherRun rF = gather_at(gF, cnt - 1, blk), rGr = gather_at(gGr, cnt - 1, blk), r3 = gather_at(g3, cnt - 1, blk), r4 = gather_at(g4, cnt - 1, blk),
                      r5 = gather_at(g5, cnt - 1, blk);
            // operand tiles of step tl-1 are gathered while step tl waits at its mailboxes (the pool is read-only inside the chunk)
            d4 Ft = gather_run(rF, -1), Grep = gather_run(rGr, -1), O3 = waveA ? gather_run<1>(r3, -1) : gather_run(r3, -1), O4 = gather_run(r4, -1),
               O5 = waveA ? z : gather_run<1>(r5, -1);
            for (int tl = cnt - 1; tl >= 0; --tl) {
                const int t = t0 + tl;
                F2_FINE(0, t == 20);
                d4 Pn = z, PY2 = z, Pm = z;
                double Qux0 = 0.0, IK0 = 0.0;
                const d4 Fc = Ft, Gc = Grep, C3 = O3, C4 = O4, C5 = O5;
                if (waveA) {
                    const d4 PF = mma_tn(P, Fc, z);                     // P F
                    Qux0 = mma4_tn(Gc, PF, C3[0]);                      // Qux = Hux + G' P F
                    Pn = mma_tn(Fc, PF, C4);                            // Hxx + F' P F
                    Qx[lane] = Qux0;
                } else {                                                // the m x m solve is the critical path of a step: nothing else in front of it
                    PY2 = mma_tn(P, C3, W2);                            // [P G | P E + W]
                    const double Q20 = mma4_tn(Gc, PY2, C5[0]);         // [Quu | Que]
                    F2_FINE(6, t == 20);
                    const double Zrep = quu_inverse_rep<M>(Q20, scratch, lane, ok);
                    F2_FINE(7, t == 20);
                    IK0 = mma4_blk(Zrep, Q20, 0.0);                     // [I | k]
                    Zx[lane] = Zrep;
                }
                F2_FINE(1, t == 20);
                wg_sync();
                F2_FINE(2, t == 20);
                if (waveA) {
                    const double Zrep = Zx[lane];
                    d4 Qux = z, K = z;
                    Qux[0] = Qux0;
                    K[0] = mma4_blk(Zrep, Qux0, 0.0);                   // K = Quu^-1 Qux
                    Pm = mms_tn_r0(Qux, K, Pn);                         // Hxx + F'PF - Qux'K
                    tile_to_lds17(Pbuf, Pm, lane);
                    store_all<1>(gw + t * GSZ, mK, K);
                } else {
                    const d4 FY = mma_tn(Fc, PY2, C4);                  // [Qux' | Wn]   (off the path to Z: after the mailbox)
                    d4 Qux = z, IK = z;
                    Qux[0] = Qx[lane];
                    IK[0] = IK0;
                    const d4 Wn = mms_tn_r0(Qux, IK, FY);
                    W2 = keep_cols(Wn, M, M + NP, lane);                // (control columns: zero up to rounding, masked - see pdp_riccati.h)
                    IK = keep_cols(IK, M, M + NP, lane);
                    store_all<1>(gw + t * GSZ + NX * NU, mIK, IK);
                }
                if (tl > 0) {                                           // next step's operands (no LDS read below the first pool row)
                    Ft = gather_run(rF, -1); Grep = gather_run(rGr, -1); O4 = gather_run(r4, -1);
                    if (waveA) O3 = gather_run<1>(r3, -1); else { O3 = gather_run(r3, -1); O5 = gather_run<1>(r5, -1); }
                }
                F2_FINE(3, t == 20);
                wg_sync();
                F2_FINE(4, t == 20);
                // both waves: P <- (P + P')/2   (wave A still holds P- in registers and reads only the transpose back)
                P = 0.5 * ((waveA ? Pm : tile_from_lds17(Pbuf, lane)) + tile_from_lds17_transposed(Pbuf, lane));
                F2_FINE(5, t == 20);
            }
            F2_ACC(2, tc2);
        }
    }
    F2_STAMP();
#ifdef PDP_PHASE_TIMING
    if (!waveA && lane == 0 && b == 0) { long long* o = (long long*)(loss + B) + 32; for (int i = 0; i < 8; ++i) o[i] = fs[i]; }
#endif
    bool finite = tile_finite(P) && tile_finite(W2);
    if (!waveA && lane == 0) { fl[0] = ok ? 1.0 : 0.0; }
    if (!waveA) { const bool f = __all(finite); if (lane == 0) fl[1] = f ? 1.0 : 0.0; }
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // gains written by both waves are read by wave A below
    wg_sync();
    if (!waveA) return;                                 // the forward sweep is one serial chain: wave A

    // ---------------- forward sweep (wave A): sensitivities, loss and gradient ------------------------------------------------------
    ok = ok && fl[0] != 0.0;
    finite = finite && fl[1] != 0.0;
    double acc = 0.0, lsum = 0.0;
    {
        wave_lds_sync();
        for (int i_ = lane; i_ < Mdl::FWD_NCONST; i_ += 64) blk[1 + i_] = Mdl::fwd_const(i_);
        constexpr int DLX = Mdl::FWD_NVAR, DLU = Mdl::FWD_NVAR + NX;
        Gather gFT, gGT, gE, gDX, gDU;
        make_gather(gFT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX && c < NX) ? Mdl::fwd_code(0, c * NX + r) : -1; });
        make_gather(gGT, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < M && c < NX) ? Mdl::fwd_code(1, c * NU + r) : -1; });
        make_gather(gE, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fwd_code(2, r * NP + (c - M)) : -1; });
        make_gather(gDX, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < NX) ? DLX + r : -1; });
        make_gather(gDU, lane, L::NC, L::FSTRIDE, [](int r, int c) { return (r < M) ? DLU + r : -1; });
        const double* dxb = demo_x + (int64_t)b * (T + 1) * NX;
        const double* dub = demo_u + (int64_t)b * T * NU;
        d4 X2 = z;
        const TileMapBytes mKT = to_bytes_sink(make_rep4_map_transposed(NX, NU, NX, lane), fused_gain0_doubles<Mdl>() - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
        d4 KTn = -load_all<4>(gw, mKT);
        d4 kn = -load_all<1>(gw + NX * NU, mIK);
        const int nchunk = (T + CH - 1) / CH;
        const int ch = (T + nchunk - 1) / nchunk;
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ch, cnt = min(ch, T - t0);
            wave_lds_sync();
            if (lane < cnt) {
                PDP_F2_PAR();
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
            GatherRun rFT = gather_at(gFT, 0, blk), rGT = gather_at(gGT, 0, blk), rE = gather_at(gE, 0, blk), rDX = gather_at(gDX, 0, blk), rDU = gather_at(gDU, 0, blk);
            for (int tl = 0; tl < cnt; ++tl) {
                const int t = t0 + tl, tnx = (t + 1 < T) ? t + 1 : t;
                const d4 KTc = KTn, kc = kn;
                KTn = -load_all<4>(gw + tnx * GSZ, mKT);
                kn = -load_all<1>(gw + tnx * GSZ + NX * NU, mIK);
                const d4 FT = gather_run(rFT, 1), GT = gather_run<1>(rGT, 1), E2 = gather_run(rE, 1), DX = gather_run(rDX, 1), DU = gather_run<1>(rDU, 1);
                d4 U2, Xn;
                riccati_forward(KTc, kc, FT, GT, E2, X2, U2, Xn);
                acc += DX[0] * X2[0] + DX[1] * X2[1] + DX[2] * X2[2] + DX[3] * X2[3] + DU[0] * U2[0];
                if (dxdp) store_dense(dxdp + ((int64_t)b * (T + 1) + t) * NX * NP, NX, NP, NP, 0, M, lane, X2);
                if (dudp) store_dense(dudp + ((int64_t)b * T + t) * NU * NP, NU, NP, NP, 0, M, lane, U2);
                X2 = Xn;
            }
        }
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
    const int gstride = (flags & PDP_OC_PACKED) ? NP + 1 : NP;
    if (lane >= M && lane < M + NP) grad[(int64_t)b * gstride + (lane - M)] = acc;
    if (lane == 0) { loss[b] = lsum; if (flags & PDP_OC_PACKED) grad[(int64_t)b * gstride + NP] = lsum; }
    int st = 0;
    if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
    if (!ok) st |= PDP_STATUS_PIVOT;
    if (lane == 0 && status) status[b] = st;
#ifdef PDP_PHASE_TIMING
    F2_STAMP();
    if (lane == 0 && b == 0) {
        long long* o = (long long*)(loss + B);
        for (int i = 0; i < 8; ++i) o[i] = ts[i];
        for (int i = 0; i < 8; ++i) o[8 + i] = fs[i];
        for (int i = 0; i < 4; ++i) o[16 + i] = acc_t[i];
    }
#endif
#undef PDP_F2_PAR
}

}  // namespace pdp
