// pdp_ocsolve2_kernels.h - the multiple-shooting optimal-control solve (OCSys.ocSolver, reference PDP/PDP.py:121-220) as a RUNNER / EVALUATOR
// pair of wavefronts per trajectory: oc_solve_ms2_kernel.  Same NLP, same iteration (IPOPT's: least-squares initial multipliers, primal-dual
// Newton step, inertia correction, filter line search - see pdp_ocsolve_kernels.h, whose one-wave kernel this replaces as the default), same
// inputs / outputs / status bits; oracle/ipopt_ms.py is the CPU restatement both are tested against.
//
// Why.  One Newton iteration of the one-wave kernel at C3 (quadrotor, T = 50) took 308 k cycles (profiles/r03_ms_phase_timing_before.txt):
// Riccati steps 142 k (2830 per step: 256 VGPRs + 228 AGPRs, the evaluation code and the MFMA chains in one register allocation), forward steps
// 84 k (1670 per step: the gains AND P_{t+1}, W_{t+1} - 1.9 KB per stage, 98 MB per sweep at B = 1024 - came back from HBM one step ahead,
// which does not cover the latency), evaluation passes / residual pass / line search 67 k, all of it serial in one instruction stream.
// Here, as in pdp_fused3_kernels.h, the work is split by KIND between two waves that share the trajectory's LDS slice:
//     runner     the master: IPOPT's control flow and the two serial MFMA chains - the Riccati backward steps and the forward steps
//                (dx, du only) - over pools of KKT-matrix entries it finds ready in LDS;
//     evaluator  everything that is lane-per-stage: the KKT matrices of a chunk of stages one chunk AHEAD of the runner (double-buffered
//                pool), the residuals of every trial point (defects, Lagrangian gradients, objective - ONE pass gives the line search its
//                (theta, phi) and, when the point is accepted, the next iteration its convergence test and right-hand sides: no separate
//                residual pass, no dyn / costate / H_u re-evaluation inside the sweep), the multiplier step dlam_t = P_{t+1} dx_{t+1} +
//                W_{t+1} with one lane per stage behind the runner's forward chunks (169 FMAs per lane instead of a 256-cycle MFMA product
//                on the runner's chain, and P never comes back to the runner), the update of (x, u, lambda).
// The runner sends commands (sweep / trial / update / exit) through a mailbox in LDS; chunks are handed over with produced / consumed
// counters (release / acquire at workgroup scope); every wait has a watchdog (a protocol error ends the trajectory with PDP_MS_INTERNAL
// instead of hanging the GPU).
//
// Placement.  TPW trajectories per workgroup of 2 TPW waves, runner = wave j, evaluator = wave j + TPW.  TPW = 4 (512 threads): the pair
// shares a SIMD (probes/wave_placement_probe.hip) - 1024 trajectories fill the chip with every SIMD running one runner and its
// evaluator.  TPW = 1 / 2 (small batches, e.g. C2's 256 problems or a 512-trajectory shard): the pair sits on two different SIMDs of
// the CU, nothing is shared but the LDS.
//
// Workspace traffic.  Per stage the sweep leaves K [m x n], k [m], W_{t+1} [n] and P_{t+1} - for n > 4 only its upper triangle (P is
// symmetrised every step, so the two halves are bit-identical): 160 instead of 238 doubles for the quadrotor.  The runner's forward
// steps re-read only K, k (56 doubles), requested TWO steps ahead through buffer loads (absent tile elements are out-of-range lanes: they
// load 0 and store nothing, no sink words, no predicated blocks).
#pragma once
#include "pdp_ocsolve_kernels.h"
#include "pdp_fused3_kernels.h"

namespace pdp {

template <class Mdl>
struct Ms2Layout {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    static constexpr bool SMALL = NX <= 4;
    static constexpr int NS = Mdl::SOL_NVAR, NF = Mdl::SOLF_NVAR;
    // backward pool row: [sol entries | defect c_t (NX) | grad_x L (NX) | grad_u L (NU) | 0.0 | constants]  (uniform rows, see Fused3Layout)
    static constexpr int C0 = NS, RX = NS + NX, RU = NS + 2 * NX, CB0 = NS + 2 * NX + NU;
    static constexpr int BSTRIDE = (CB0 + 1 + Mdl::SOL_NCONST) | 1;
    // forward pool row: [solf entries | defect c_t (NX) | grad_x L of stage t+1 (NX) | grad_u L (NU) | 0.0 | constants]
    static constexpr int FC0 = NF, FRX = NF + NX, FRU = NF + 2 * NX, CF0 = NF + 2 * NX + NU;
    static constexpr int FSTRIDE = (CF0 + 1 + Mdl::SOLF_NCONST) | 1;
    // a trajectory's LDS slice (doubles)
    static constexpr int FIN = RICCATI_SCRATCH;                          // [0.0 | terminal constants | terminal entries]
    static constexpr int NCFIN = 1 + Mdl::FIN_NCONST;
    static constexpr int PAR = FIN + NCFIN + Mdl::FIN_NVAR;               // theta (NP) | theta-only precomputed values (NPC)
    static constexpr int DLT = PAR + NP + Mdl::NPC;                       // NX: terminal gradient h_x(x_T) - lambda_T
    static constexpr int CTL = (DLT + NX + 1) & ~1;                       // mailbox: 16 ints | 24 doubles
    static constexpr int POOL = CTL + 32;
    static constexpr int SLICE = 160 * 1024 / 8 / 4;
    static constexpr int BUF = (SLICE - POOL) / 2;
    static constexpr int ROWS = BUF / BSTRIDE < 64 ? BUF / BSTRIDE : 64;
    static constexpr int ROWSF = BUF / FSTRIDE < 64 ? BUF / FSTRIDE : 64;
    // workspace per stage
    static constexpr int GSZ = NX * NU + NU;                              // K [NU x NX] | k [NU]
    static constexpr int PSZ = SMALL ? NX * NX : NX * (NX + 1) / 2;      // P_{t+1}: full (small systems keep it in rep form), else the upper triangle
    static constexpr int PWSZ = PSZ + NX;                                 //          | W_{t+1} [NX]
    __host__ __device__ static constexpr int pk(int i, int j) { return SMALL ? i * NX + j : (i <= j ? i * NX - i * (i - 1) / 2 + (j - i) : j * NX - j * (j - 1) / 2 + (i - j)); }
    __host__ __device__ static constexpr int64_t res_doubles(int T) { return (int64_t)T * NX + (int64_t)(T + 1) * NX + (int64_t)T * NU; }      // c | grad_x L | grad_u L
    // workspace per trajectory (doubles): dx | du | dlam | residual set 0 | residual set 1 | gains | P, W | filter (theta, phi)
    __host__ __device__ static constexpr int64_t ws_doubles(int T, int max_iter) {
        return (int64_t)(T + 1) * NX + (int64_t)T * NU + (int64_t)T * NX + 2 * res_doubles(T) + (int64_t)T * GSZ + (int64_t)T * PWSZ + 2 * (int64_t)(max_iter + 1);
    }
};

template <class Mdl>
__host__ __device__ constexpr bool ms2_ok() {
    using L = Ms2Layout<Mdl>;
    return Mdl::NX <= 16 && Mdl::NU <= 4 && L::ROWS >= 4 && L::ROWSF >= 4 && Mdl::FIN_NVAR + L::NCFIN + L::PAR <= L::SLICE;
}

#define PDP_MS_INTERNAL 64      /* status bit: the runner / evaluator hand-over timed out (a bug, never expected) */
#define PDP_MS_NOGAINS 32       /* status bit: gains were requested but no complete positive definite sweep exists at the returned point (zeros written) */

// mailbox slots (ints) and result slots (doubles behind them)
enum { MS2_SEQ = 0, MS2_TYPE = 1, MS2_PROD = 2, MS2_CONS = 3, MS2_DONE = 4, MS2_ABORT = 5, MS2_DEAD = 6, MS2_CUR = 7, MS2_DST = 8 };
enum { MS2_CMD_EXIT = 0, MS2_CMD_SWEEP = 1, MS2_CMD_TRIAL = 2, MS2_CMD_UPDATE = 3 };
enum { MS2_ALPHA = 0, MS2_F = 1, MS2_TH = 2, MS2_PR = 3, MS2_DU = 4, MS2_Z = 5, MS2_L = 6, MS2_LC = 7, MS2_FIN = 8 };

PDP_DEV int ms2_load(int* f) { return __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// wait until *f >= v; false when the partner never gets there (watchdog) or the trajectory has been declared dead
PDP_DEV bool ms2_wait_ge(int* f, int v, int* ctl) {
    int n = 0;
    while (ms2_load(f) < v) {
        __builtin_amdgcn_s_sleep(2);
        if (++n > (1 << 22) || ms2_load(ctl + MS2_DEAD) != 0) { f3_signal(ctl + MS2_DEAD, 1); return false; }
    }
    return true;
}

// tile <-> workspace through buffer instructions: voff[r] = byte offset of element (lane, r) inside a stage's record, or out of range (the
// hardware drops such lanes of a store and returns 0 for them in a load); the stage offset travels in the scalar offset
struct BufMap { unsigned voff[4]; };
constexpr unsigned MS2_OOB = 0x80000000u;
template <int NR = 4, class RS>
PDP_DEV void buf_store(RS rs, unsigned soff, const BufMap& m, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double x = v[r];
        pdp_u2 w;
        w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x);
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, m.voff[r], soff, 0);
    }
}
template <int NR = 4, class RS>
PDP_DEV d4 buf_load(RS rs, unsigned soff, const BufMap& m) {
    d4 v = zero4();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const pdp_u2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, m.voff[r], soff, 0);
        v[r] = __hiloint2double((int)w.y, (int)w.x);
    }
    return v;
}

template <class Mdl, int TPW>
__global__ void __launch_bounds__(128 * TPW) oc_solve_ms2_kernel(int B, int T, pdp_oc_ms_opts op, const double* __restrict__ x0, const double* __restrict__ theta,
                                                                  int tb, double* __restrict__ x, double* __restrict__ u, double* __restrict__ lam,
                                                                  double* __restrict__ cost, double* __restrict__ resid, int32_t* __restrict__ converged,
                                                                  int32_t* __restrict__ iters, int32_t* __restrict__ status, double* __restrict__ gains_out,
                                                                  double* __restrict__ iter_log, double* __restrict__ ws) {
    using L = Ms2Layout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, M = NU;
    constexpr bool SMALL = L::SMALL;
    constexpr int NRT = SMALL ? 1 : 4;
    constexpr int GSZ = L::GSZ, PSZ = L::PSZ, PWSZ = L::PWSZ;
    constexpr int BS = L::BSTRIDE, FS = L::FSTRIDE;
    constexpr int U = 4;                                     // backward steps per address update (literal row offsets inside a group)
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "trajectories per workgroup");
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = wid & (TPW - 1);
    const bool runner = wid < TPW;
    const int b = blockIdx.x * TPW + slot;
    double* lds = lds_all + slot * L::SLICE;
    double* scratch = lds;
    double* fin = lds + L::FIN;
    double* par = lds + L::PAR;
    double* dlT = lds + L::DLT;
    int* ctl = (int*)(lds + L::CTL);
    double* res = lds + L::CTL + 8;                          // result / parameter slots (MS2_ALPHA ..)
    double* pool = lds + L::POOL;
    if (runner && lane < 16) ctl[lane] = 0;
    __syncthreads();                                         // the only workgroup barrier: mailbox zeroed before anyone polls it
    if (b >= B) return;
    const d4 z = zero4();
    const int tlane = small_transpose_lane(lane);
#define PDP_MS2_PAR()                                                     \
    double th[NP > 0 ? NP : 1], pc[Mdl::NPC];                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) th[i_] = par[i_];  \
    _Pragma("unroll") for (int i_ = 0; i_ < Mdl::NPC; ++i_) pc[i_] = par[NP + i_]
    double* xb = x + (int64_t)b * (T + 1) * NX;
    double* ub = u + (int64_t)b * T * NU;
    double* lb = lam + (int64_t)b * T * NX;
    double* w0 = ws + (int64_t)b * L::ws_doubles(T, op.max_iter);
    double* dxb = w0;                                        // (T+1) x NX
    double* dub = dxb + (int64_t)(T + 1) * NX;               // T x NU
    double* dlb = dub + (int64_t)T * NU;                     // T x NX
    double* rs0 = dlb + (int64_t)T * NX;                     // residual sets 0 / 1: c [T][NX] | grad_x L [T+1][NX] | grad_u L [T][NU]
    const int64_t RES = L::res_doubles(T);
    double* gw = rs0 + 2 * RES;                              // gains, T x GSZ
    double* pw = gw + (int64_t)T * GSZ;                      // P_{t+1}, W_{t+1}, T x PWSZ
    double* fth = pw + (int64_t)T * PWSZ;                    // filter: theta entries (at most one per iteration) ...
    double* fph = fth + (op.max_iter + 1);                   //         ... and phi entries
    // chunks: backward chunk g (0 = last stages) covers [t0, t0 + cnt); forward chunks follow in the same numbering
    const int nchunk = (T + L::ROWS - 1) / L::ROWS;
    const int ch = (T + nchunk - 1) / nchunk;
    auto bchunk = [&](int g, int& t0, int& cnt) { const int c = nchunk - 1 - g; t0 = c * ch; cnt = min(ch, T - t0); };
    const int nchunkF = (T + L::ROWSF - 1) / L::ROWSF;
    const int chF = (T + nchunkF - 1) / nchunkF;

    if (runner) {
        // ================================================ runner ================================================
        __builtin_amdgcn_s_setprio(3);
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
        }
        // ---- starting point: the caller's (x, u, lambda) [PDP_MS_WARM], or IPOPT's: w0 = 0 (PDP.py:155,166), x_0 = ini_state
        const bool warm = (op.flags & PDP_MS_WARM) != 0;
        if (!warm) {
            for (int i = lane; i < (T + 1) * NX; i += 64) xb[i] = i < NX ? x0[(int64_t)b * NX + i] : 0.0;
            for (int i = lane; i < T * NU; i += 64) ub[i] = 0.0;
            for (int i = lane; i < T * NX; i += 64) lb[i] = 0.0;
        } else if (lane < NX) xb[lane] = x0[(int64_t)b * NX + lane];
        for (int i = lane; i < NX; i += 64) dxb[i] = 0.0;   // x_0 is fixed
        bool dead = false;
        int seq = 0;
        // commands: parameters first, then the sequence number with release semantics (LDS writes and global stores above are visible to the evaluator)
        auto issue = [&](int type, double alpha, int cur, int dst) {
            if (lane == 0) { ctl[MS2_TYPE] = type; ctl[MS2_CUR] = cur; ctl[MS2_DST] = dst; res[MS2_ALPHA] = alpha; ctl[MS2_PROD] = 0; ctl[MS2_CONS] = 0; }
            ++seq;
            f3_signal(ctl + MS2_SEQ, seq);
        };
        auto wait_done = [&]() { if (!dead && !ms2_wait_ge(ctl + MS2_DONE, seq, ctl)) dead = true; };

        // ---- loop-invariant gather / store maps
        auto codeS = [](int mat, int i) { return Mdl::sol_code(mat, i); };       // 0 F, 1 G, 2 Hxx, 3 Hxu, 4 Huu
        Gather3 gF, gY, gHxx, gHX, gHU, gGr, gHux;
        make_gather3(gF, lane, L::CB0, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(0, r * NX + (c & 3)) : -1)
                                                                        : ((r < NX && c < NX) ? codeS(0, r * NX + c) : -1); });
        make_gather3(gY, lane, L::CB0, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(1, r * NU + c) : (c == M ? L::C0 + r : -1)); });
        make_gather3(gGr, lane, L::CB0, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeS(1, r * NU + (c & 3)) : -1; });
        make_gather3(gHux, lane, L::CB0, [&](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? codeS(3, (c & 3) * NU + r) : -1)
                                                                          : ((r < M && c < NX) ? codeS(3, c * NU + r) : -1); });
        make_gather3(gHxx, lane, L::CB0, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(2, r * NX + (c & 3)) : -1)
                                                                          : ((r < NX && c < NX) ? codeS(2, r * NX + c) : -1); });
        make_gather3(gHX, lane, L::CB0, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(3, r * NU + c) : (c == M ? L::RX + r : -1)); });
        make_gather3(gHU, lane, L::CB0, [&](int r, int c) { return r >= M ? -1 : (c < M ? codeS(4, r * NU + c) : (c == M ? L::RU + r : -1)); });
        const int col = tile_col(lane);
        BufMap mK, mIK, mP, mW, mKT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tile_row(lane, r);
            mK.voff[r] = (row < NU && col < NX) ? 8u * (unsigned)(row * NX + col) : MS2_OOB;                          // K [NU x NX] (rep form: first column block)
            mIK.voff[r] = (row < NU && col == M) ? 8u * (unsigned)(NX * NU + row) : MS2_OOB;                          // k behind it
            mP.voff[r] = (row < NX && col < NX && (SMALL || row <= col)) ? 8u * (unsigned)L::pk(row, col) : MS2_OOB;   // P_{t+1}: full / upper triangle
            mW.voff[r] = (row < NX && col == M) ? 8u * (unsigned)(PSZ + row) : MS2_OOB;
            mKT.voff[r] = (row < NX && (col & 3) < NU) ? 8u * (unsigned)((col & 3) * NX + row) : MS2_OOB;             // K read back transposed, replicated in the column blocks
        }
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, (int)((int64_t)T * GSZ * 8), 0x00020000);
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)pw, 0, (int)((int64_t)T * PWSZ * 8), 0x00020000);
        // per-lane tile masks: diagonal of the n x n / m x m blocks
        d4 dgN, dgM0 = z;
#pragma unroll
        for (int r = 0; r < 4; ++r) dgN[r] = SMALL ? ((r == 0 && (lane >> 4) == (col & 3) && (col & 3) < NX) ? 1.0 : 0.0) : ((tile_row(lane, r) == col && col < NX) ? 1.0 : 0.0);
        dgM0[0] = ((lane >> 4) == col && col < M) ? 1.0 : 0.0;

        // residuals of the current iterate, as the evaluator's last accepted pass left them
        double f_cur = 0.0, th_cur = 0.0, inf_pr = 0.0, inf_du = 0.0, zmax = 0.0, lmax = 0.0, lamc = 0.0;
        bool finite = true;
        auto read_res = [&]() {
            wave_lds_sync();
            f_cur = res[MS2_F]; th_cur = res[MS2_TH]; inf_pr = res[MS2_PR]; inf_du = res[MS2_DU]; zmax = res[MS2_Z]; lmax = res[MS2_L]; lamc = res[MS2_LC];
            finite = res[MS2_FIN] != 0.0;
        };
        bool PWfinite = true;

        // Backward sweep over the chunks of the current SWEEP command with Hessian scale hs (1; 0 = least-squares multiplier estimate: W = I,
        // no defects) and shift dw.  Returns true when every Quu was positive definite; stops at the first one that is not.
        auto backward = [&](double hs, double dw) -> bool {
            const bool scaled = !(hs == 1.0 && dw == 0.0);
            const double sU = col < M ? hs : 1.0, sC = col == M ? hs : 1.0;      // scale of the Hessian columns / of the defect column
            bool pdall = true, ok = true;
            d4 P = z, W2 = z;
            constexpr int RB = 8 * BS;
            for (int g = 0; g < nchunk && pdall && !dead; ++g) {
                int t0, cnt;
                bchunk(g, t0, cnt);
                const double* pb = pool + (g & 1) * L::BUF;
                if (!ms2_wait_ge(ctl + MS2_PROD, g + 1, ctl)) { dead = true; break; }
                if (g == 0) {       // terminal stage (the evaluator filled it before the first chunk): P = hs hxx + dw I, W = h_x(x_T) - lambda_T
                    Gather gP;
                    make_gather(gP, lane, L::NCFIN, 0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::fin_code(0, r * NX + (c & 3)) : -1)
                                                                                       : ((r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1); });
                    P = gather_tile(fin, gP, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = tile_row(lane, r);
                        P[r] = hs * P[r] + dw * dgN[r];
                        if (col == M && row < NX) W2[r] = dlT[row];
                    }
                }
                int tl = cnt - 1;
                const double* r0 = pb + (tl - 1) * BS;      // the runs sit one row BELOW the step's row: the step reads at +RB, its one-step-ahead requests at +0
                Run3 rF = run3_at(gF, r0), rY = run3_at(gY, r0), rHxx = run3_at(gHxx, r0), rHX = run3_at(gHX, r0), rHU = run3_at(gHU, r0),
                     rGr = run3_at(gGr, r0), rHux = run3_at(gHux, r0);
                auto move_all = [&](int bytes) {
                    move3<NRT>(rF, bytes); move3<NRT>(rY, bytes); move3<NRT>(rHxx, bytes); move3<NRT>(rHX, bytes); move3<1>(rHU, bytes); move3<NRT>(rGr, bytes); move3<1>(rHux, bytes);
                };
                d4 Fa = read3<NRT>(rF, RB), Ya = read3<NRT>(rY, RB), Fb = z, Yb = z;
                auto bstep = [&](int tl, unsigned imm, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {
                    const int t = t0 + tl;
                    d4 Hxx = read3<NRT>(rHxx, imm), HX2 = read3<NRT>(rHX, imm), HU2 = read3<1>(rHU, imm), Grep = read3<NRT>(rGr, imm), Hux = read3<1>(rHux, imm);
                    if (tl > 0) { Fn = read3<NRT>(rF, imm - RB); Yn = read3<NRT>(rY, imm - RB); }
                    d4 Ys = Yc;
                    double Hux0 = Hux[0];
                    if (scaled) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { Hxx[r] = hs * Hxx[r] + dw * dgN[r]; HX2[r] *= sU; }
                        HU2[0] = sU * HU2[0] + dw * dgM0[0];
                        Ys = Yc * sC;
                        Hux0 = hs * Hux0;
                    }
                    // P_{t+1}, W_{t+1}: the evaluator's multiplier step needs them (dlam_t = P_{t+1} dx_{t+1} + W_{t+1})
                    const unsigned soP = (unsigned)(t * PWSZ) * 8u, soG = (unsigned)(t * GSZ) * 8u;
                    buf_store<NRT>(rsP, soP, mP, P);
                    buf_store<NRT>(rsP, soP, mW, W2);
                    if constexpr (SMALL) {
                        SmallGains gs;
                        double Pr = P[0], Wr = W2[0];
                        ok = riccati_small_backward<M>(Pr, Wr, Fc[0], Ys[0], Grep[0], Hxx[0], HX2[0], HU2[0], Hux0, lane, tlane, 1, gs) && ok;
                        P[0] = Pr; W2[0] = Wr;
                        pdall = pdall && gs.pd;
                        d4 Kt = z, IKt = z;
                        Kt[0] = gs.K; IKt[0] = gs.IK;
                        buf_store<1>(rsG, soG, mK, Kt);
                        buf_store<1>(rsG, soG, mIK, IKt);
                    } else {
                        RiccatiGains gn;
                        d4 P_old;
                        ok = riccati_backward<M, false, false, true>(P, W2, Fc, Ys, Grep, Hxx, HX2, HU2, Hux0, scratch, lane, 1, gn, P_old) && ok;
                        pdall = pdall && gn.pd;
                        buf_store<1>(rsG, soG, mK, gn.K);
                        buf_store<1>(rsG, soG, mIK, gn.IK);
                    }
                };
                // single steps until a whole number of groups of U remains, then groups of U steps with literal row offsets
                for (; (tl + 1) % U != 0 && pdall; --tl) { bstep(tl, RB, Fa, Ya, Fb, Yb); Fa = Fb; Ya = Yb; move_all(-RB); }
                move_all(-(U - 1) * RB);
                for (; tl >= U - 1 && pdall; tl -= U) {
                    bstep(tl, (unsigned)(U * RB), Fa, Ya, Fb, Yb);
                    if (pdall) bstep(tl - 1, (unsigned)((U - 1) * RB), Fb, Yb, Fa, Ya);
                    if (pdall) bstep(tl - 2, (unsigned)((U - 2) * RB), Fa, Ya, Fb, Yb);
                    if (pdall) bstep(tl - 3, (unsigned)((U - 3) * RB), Fb, Yb, Fa, Ya);
                    move_all(-U * RB);
                }
                if (pdall) f3_signal(ctl + MS2_CONS, g + 1);
            }
            pdall = pdall && ok && !dead;
            PWfinite = pdall ? __all(tile_finite(P) && tile_finite(W2)) : true;      // (an aborted sweep leaves P, W undefined)
            return pdall;
        };

        // Forward pass of the LQ problem over the forward chunks: dx, du into the workspace (the evaluator follows with dlam);
        // returns grad(phi)' d = grad(L)' d + lambda' c  (A d = -c)
        auto forward = [&](double hs) -> double {
            Gather3 gFT, gGT, gE, gRX, gRU;
            make_gather3(gFT, lane, L::CF0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::solf_code(0, (c & 3) * NX + r) : -1)
                                                                          : ((r < NX && c < NX) ? Mdl::solf_code(0, c * NX + r) : -1); });
            make_gather3(gGT, lane, L::CF0, [](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? Mdl::solf_code(1, (c & 3) * NU + r) : -1)
                                                                          : ((r < M && c < NX) ? Mdl::solf_code(1, c * NU + r) : -1); });
            make_gather3(gE, lane, L::CF0, [](int r, int c) { return (r < NX && c == M) ? L::FC0 + r : -1; });
            make_gather3(gRX, lane, L::CF0, [](int r, int c) { return (r < NX && c == M) ? L::FRX + r : -1; });
            make_gather3(gRU, lane, L::CF0, [](int r, int c) { return (r < M && c == M) ? L::FRU + r : -1; });
            const ColStore csX = make_col_store(NX, M, lane), csU = make_col_store(NU, M, lane);
            const auto rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dxb, 0, (int)(((int64_t)(T + 1) * NX + (int64_t)T * NU) * 8), 0x00020000);
            const unsigned offU = (unsigned)((T + 1) * NX) * 8u;
            // feedback gains of stage t are requested two steps ahead (three register sets in rotation)
            struct Gn { d4 KT, k; };
            auto ldg = [&](int t) { const int tt = t < T ? t : T - 1; Gn s; const unsigned so = (unsigned)(tt * GSZ) * 8u; s.KT = -buf_load<NRT>(rsG, so, mKT); s.k = -buf_load<1>(rsG, so, mIK); return s; };
            Gn A = ldg(0), Bn = ldg(1), Cn;
            Cn.KT = z; Cn.k = z;
            d4 X2 = z, Xb = z;
            double acc = 0.0;
            const bool scaledE = hs != 1.0;
            constexpr int RF = 8 * FS;
            for (int c = 0; c < nchunkF && !dead; ++c) {
                const int g = nchunk + c, t0 = c * chF, cnt = min(chF, T - t0);
                const double* pb = pool + (g & 1) * L::BUF;
                if (!ms2_wait_ge(ctl + MS2_PROD, g + 1, ctl)) { dead = true; break; }
                Run3 rFT = run3_at(gFT, pb), rGT = run3_at(gGT, pb), rE = run3_at(gE, pb), rRX = run3_at(gRX, pb), rRU = run3_at(gRU, pb);
                auto move_all = [&](int bytes) { move3<NRT>(rFT, bytes); move3<1>(rGT, bytes); move3<NRT>(rE, bytes); move3<NRT>(rRX, bytes); move3<1>(rRU, bytes); };
                auto fstep = [&](int tl, unsigned imm, const d4 Xc, d4& Xn, const Gn& cur, Gn& fill) {
                    const int t = t0 + tl;
                    fill = ldg(t + 2);
                    d4 FT = read3<NRT>(rFT, imm);
                    d4 GT = read3<1>(rGT, imm);
                    d4 E2 = read3<NRT>(rE, imm);
                    d4 RXn = read3<NRT>(rRX, imm);
                    d4 RUc = read3<1>(rRU, imm);
                    if (scaledE) E2 = E2 * hs;
                    d4 U2;
                    if constexpr (SMALL) {
                        U2 = z; Xn = z;
                        double u0, x1;
                        riccati_small_forward(cur.KT[0], cur.k[0], FT[0], GT[0], E2[0], Xc[0], u0, x1);
                        U2[0] = u0; Xn[0] = x1;
                        acc += RXn[0] * x1 + RUc[0] * u0;
                    } else {
                        riccati_forward(cur.KT, cur.k, FT, GT, E2, Xc, U2, Xn);
                        acc += RXn[0] * Xn[0] + RXn[1] * Xn[1] + RXn[2] * Xn[2] + RXn[3] * Xn[3] + RUc[0] * U2[0];
                    }
                    store_tile_column_buf<1>(rsD, offU + (unsigned)(t * NU) * 8u, csU, U2);
                    store_tile_column_buf<NRT>(rsD, (unsigned)((t + 1) * NX) * 8u, csX, Xn);
                };
                int tl = 0;
                for (; tl + 3 <= cnt; tl += 3) {
                    fstep(tl, 0u, X2, Xb, A, Cn);
                    fstep(tl + 1, (unsigned)RF, Xb, X2, Bn, A);
                    fstep(tl + 2, (unsigned)(2 * RF), X2, Xb, Cn, Bn);
                    X2 = Xb;
                    move_all(3 * RF);
                }
                for (; tl < cnt; ++tl) { fstep(tl, 0u, X2, Xb, A, Cn); X2 = Xb; A = Bn; Bn = Cn; move_all(RF); }
                f3_signal(ctl + MS2_CONS, g + 1);       // release: dx, du of the chunk are in memory
            }
            return wave_sum(acc) + lamc;
        };

        // ---- main loop (IPOPT's order: convergence test, sweep with inertia correction, line search).
        // phase 0 (cold start only): the least-squares multiplier estimate (constr_mult_init_max = 1000),
        //     [I A'; A 0] [w; lambda] = -[grad f; 0]  - the same sweep with W = I and no defects; phase 1: the iteration.
        int st = 0, it = 0, nfilt = 0, conv = 0, phase = warm ? 1 : 0, cur = 0;
        double hs = warm ? 1.0 : 0.0, dw = warm ? 0.0 : 1.0, dw_last = 0.0, theta_max = 0.0, theta_min = 0.0;
        bool gains_ok = false;
        // residuals of the starting point (both phases: the phase-0 sweep takes its right-hand sides from the same arrays)
        issue(MS2_CMD_TRIAL, 0.0, cur, cur);
        wait_done();
        read_res();
        for (;;) {
            if (dead) break;
            if (phase == 1 && dw == 0.0) {              // a new iterate: converged?  (a sweep follows only if not - or once more for the gains output)
                if (!finite) { st |= PDP_STATUS_NONFINITE; break; }
                if (it == 0) { theta_max = 1e4 * fmax(1.0, th_cur); theta_min = 1e-4 * fmax(1.0, th_cur); }
                if (inf_pr <= op.tol * (1.0 + zmax) && inf_du <= op.tol * (1.0 + lmax)) { conv = 1; if (!gains_out) break; }
                else if (it >= op.max_iter) { st |= PDP_MS_MAXITER; break; }
            }
            issue(MS2_CMD_SWEEP, 0.0, cur, cur);
            const bool pd = backward(hs, dw);
            if (dead) break;
            if (conv || !pd || (phase == 1 && !PWfinite)) {         // the sweep ends here: tell the evaluator to drop the remaining chunks
                f3_signal(ctl + MS2_ABORT, seq);
                wait_done();
                if (dead) break;
            }
            gains_ok = pd && PWfinite;
            if (conv) break;                            // (the sweep at the solution left the LQR gains in the workspace)
            if (phase == 0) {
                if (!(pd && PWfinite)) {
                    if (pd) { f3_signal(ctl + MS2_ABORT, seq); wait_done(); if (dead) break; }
                    phase = 1; hs = 1.0; dw = 0.0; continue;
                }
            } else {
                if (!PWfinite) { st |= PDP_STATUS_NONFINITE; break; }
                if (!pd) {                              // Algorithm IC (defaults: 1e-4 first, x100 / x8 up, /3 down, 1e20 max)
                    if (dw == 0.0) dw = dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last * (1.0 / 3.0));
                    else dw *= dw_last == 0.0 ? 100.0 : 8.0;
                    if (dw > 1e20) { st |= PDP_MS_INERTIA; break; }
                    continue;
                }
                if (dw > 0.0) dw_last = dw;
            }
            const double gd = forward(hs);
            wait_done();                                // the evaluator has finished dlam
            if (dead) break;
            if (phase == 0) {
                double lm = 0.0;
                bool fin = true;
                for (int q = lane; q < T * NX; q += 64) { const double v = dlb[q]; lm = fmax(lm, fabs(v)); fin = fin && fabs(v) <= 1.7e308; }
                lm = wave_max(lm);
                if (__all(fin) && lm <= 1000.0) {
                    for (int q = lane; q < T * NX; q += 64) lb[q] = dlb[q];
                    issue(MS2_CMD_TRIAL, 0.0, cur, cur);        // the residuals change with the multipliers
                    wait_done();
                    read_res();
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
            while (alpha >= amin && alpha > 1e-300) {      // (the second bound only guards against amin = 0)
                issue(MS2_CMD_TRIAL, alpha, cur, cur ^ 1);
                wait_done();
                if (dead) break;
                wave_lds_sync();
                ft = res[MS2_F]; tht = res[MS2_TH];
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
            if (dead) break;
            if (iter_log && it < op.log_rows && lane == 0) {
                double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
                row[0] = it; row[1] = f; row[2] = inf_pr; row[3] = inf_du; row[4] = dw; row[5] = accepted ? alpha : 0.0; row[6] = gd; row[7] = theta;
            }
            if (!accepted) { st |= PDP_MS_RESTORATION; break; }
            if (!ftype) {                               // (at most one entry per iteration: the workspace holds max_iter + 1)
                if (lane == 0) { fth[nfilt] = (1.0 - 1e-5) * theta; fph[nfilt] = f - 1e-8 * theta; }
                ++nfilt;
                __threadfence_block();
            }
            read_res();                                 // the accepted trial's residuals are the new iterate's
            cur ^= 1;
            issue(MS2_CMD_UPDATE, alpha, cur, cur);     // (x, u, lambda) += alpha (dx, du, dlam)
            wait_done();
            gains_ok = false;
            dw = 0.0;
            ++it;
        }
        issue(MS2_CMD_EXIT, 0.0, cur, cur);
        if (dead) st |= PDP_MS_INTERNAL;
        if (lane == 0) {
            if (cost) cost[b] = f_cur;
            if (resid) { resid[2 * b] = inf_pr; resid[2 * b + 1] = inf_du; }
            if (converged) converged[b] = conv && !dead;
            if (iters) iters[b] = it;
        }
        if (gains_out) {        // LQR feedback around the last linearisation point, in the layout of pdp_oc_rollout_feedback_batched: K^T [n][m] | k [m]
            __threadfence_block();
            constexpr int G2 = NX * NU + NU;
            double* go = gains_out + (int64_t)b * T * G2;
            const bool have = gains_ok && conv && !dead;
            for (int q = lane; q < T * G2; q += 64) {
                const int t = q / G2, r = q - t * G2;
                go[q] = !have ? 0.0 : (r < NX * NU ? gw[t * GSZ + (r % NU) * NX + r / NU] : gw[t * GSZ + r]);
            }
            if (!have) st |= PDP_MS_NOGAINS;
        }
        if (lane == 0 && status) status[b] = st;
    } else {
        // ============================================== evaluator ==============================================
        __builtin_amdgcn_s_setprio(0);
        bool dead = false;
        int last = 0;
        double a_f, a_th, a_pr, a_du, a_z, a_l, a_lc;
        bool fin_all;
        // Residuals of the point (x, u, lambda) + a (dx, du, dlam), lane = stage: defects, Lagrangian gradients (into residual set `dst`), objective,
        // constraint violation, the convergence measures.  a = 0: the current point (dx, du, dlam are not read into the result).
        auto trial = [&](double a, double* rd) {
            PDP_MS2_PAR();
            const bool stepped = a != 0.0;
            double* rc = rd;                                 // c [T][NX]
            double* rx = rd + (int64_t)T * NX;               // grad_x L [T+1][NX]  (row 0: x_0 is fixed)
            double* ru = rx + (int64_t)(T + 1) * NX;         // grad_u L [T][NU]
            a_f = 0.0; a_th = 0.0; a_pr = 0.0; a_du = 0.0; a_z = 0.0; a_l = 0.0; a_lc = 0.0;
            bool fin = true;
            for (int t = lane; t < T; t += 64) {
                double xc[NX], uc[NU], lc[NX], xn[NX], v[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double xa = xb[t * NX + i], xd = dxb[t * NX + i], xna = xb[(t + 1) * NX + i], xnd = dxb[(t + 1) * NX + i];
                    const double la = lb[t * NX + i], ld = dlb[t * NX + i];
                    xc[i] = stepped ? fma(a, xd, xa) : xa;
                    xn[i] = stepped ? fma(a, xnd, xna) : xna;
                    lc[i] = stepped ? fma(a, ld, la) : la;
                    a_z = fmax(a_z, fabs(xc[i])); a_l = fmax(a_l, fabs(lc[i]));
                }
#pragma unroll
                for (int i = 0; i < NU; ++i) { const double ua = ub[t * NU + i], ud = dub[t * NU + i]; uc[i] = stepped ? fma(a, ud, ua) : ua; a_z = fmax(a_z, fabs(uc[i])); }
                Mdl::dyn(xc, uc, th, pc, v);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double ci = v[i] - xn[i];
                    rc[t * NX + i] = ci;
                    a_th += fabs(ci); a_pr = fmax(a_pr, fabs(ci)); a_lc += lc[i] * ci;
                    fin = fin && fabs(ci) <= 1.7e308;
                }
                Mdl::costate_step(xc, uc, lc, th, pc, v);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    double g = 0.0;                          // x_0 is fixed: no stationarity row
                    if (t > 0) { const double pa = lb[(t - 1) * NX + i], pd = dlb[(t - 1) * NX + i]; g = v[i] - (stepped ? fma(a, pd, pa) : pa); }
                    rx[t * NX + i] = g;
                    a_du = fmax(a_du, fabs(g));
                    fin = fin && fabs(g) <= 1.7e308;
                }
                double hu[NU];
                Mdl::dHu(xc, uc, lc, th, pc, hu);
#pragma unroll
                for (int i = 0; i < NU; ++i) { ru[t * NU + i] = hu[i]; a_du = fmax(a_du, fabs(hu[i])); fin = fin && fabs(hu[i]) <= 1.7e308; }
                a_f += Mdl::path_cost(xc, uc, th, pc);
                if (t == T - 1) {
                    double hx[NX];
                    Mdl::dhx(xn, th, pc, hx);
                    a_f += Mdl::final_cost(xn, th, pc);
#pragma unroll
                    for (int i = 0; i < NX; ++i) {
                        const double g = hx[i] - lc[i];
                        rx[T * NX + i] = g;
                        a_du = fmax(a_du, fabs(g)); a_z = fmax(a_z, fabs(xn[i]));
                        fin = fin && fabs(g) <= 1.7e308;
                    }
                }
            }
            a_f = wave_sum(a_f); a_th = wave_sum(a_th); a_lc = wave_sum(a_lc);
            a_pr = wave_max(a_pr); a_du = wave_max(a_du); a_z = wave_max(a_z); a_l = wave_max(a_l);
            fin_all = __all(fin);
            if (lane == 0) {
                res[MS2_F] = a_f; res[MS2_TH] = a_th; res[MS2_PR] = a_pr; res[MS2_DU] = a_du; res[MS2_Z] = a_z; res[MS2_L] = a_l; res[MS2_LC] = a_lc;
                res[MS2_FIN] = fin_all ? 1.0 : 0.0;
            }
        };
        // multiplier step of the stages [t0, t0 + cnt): dlam_t = P_{t+1} dx_{t+1} + W_{t+1}   (PDP.py:604), lane = stage
        auto dlam_chunk = [&](int t0, int cnt) {
            if (lane < cnt) {
                const int t = t0 + lane;
                const double* pp = pw + (int64_t)t * PWSZ;
                double d[NX], acc[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) { d[i] = dxb[(t + 1) * NX + i]; acc[i] = pp[PSZ + i]; }
#pragma unroll
                for (int j = 0; j < NX; ++j)
#pragma unroll
                    for (int i = 0; i < NX; ++i) acc[i] = fma(pp[L::pk(i, j)], d[j], acc[i]);
#pragma unroll
                for (int i = 0; i < NX; ++i) dlb[t * NX + i] = acc[i];
            }
        };
        for (;;) {
            if (!ms2_wait_ge(ctl + MS2_SEQ, last + 1, ctl)) break;
            last = ms2_load(ctl + MS2_SEQ);
            wave_lds_sync();
            const int type = ctl[MS2_TYPE], cur = ctl[MS2_CUR], dst = ctl[MS2_DST];
            const double alpha = res[MS2_ALPHA];
            if (type == MS2_CMD_EXIT) break;
            if (type == MS2_CMD_TRIAL) {
                trial(alpha, rs0 + (int64_t)dst * RES);
                __threadfence_block();
            } else if (type == MS2_CMD_UPDATE) {
                for (int q = lane; q < T * NX; q += 64) { xb[NX + q] = fma(alpha, dxb[NX + q], xb[NX + q]); lb[q] = fma(alpha, dlb[q], lb[q]); }
                for (int q = lane; q < T * NU; q += 64) ub[q] = fma(alpha, dub[q], ub[q]);
                __threadfence_block();
            } else {        // MS2_CMD_SWEEP
                const double* rd = rs0 + (int64_t)cur * RES;
                const double* rc = rd;
                const double* rx = rd + (int64_t)T * NX;
                const double* ru = rx + (int64_t)(T + 1) * NX;
                bool aborted = false;
                auto stop = [&]() { aborted = aborted || ms2_load(ctl + MS2_ABORT) == last; return aborted || dead; };
                // terminal stage: hxx(x_T) entries and the terminal gradient
                if (lane == 0) fin[0] = 0.0;
                for (int i = lane; i < Mdl::FIN_NCONST; i += 64) fin[1 + i] = Mdl::fin_const(i);
                if (lane == 0) {
                    PDP_MS2_PAR();
                    double xT[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) { xT[i] = xb[T * NX + i]; dlT[i] = rx[T * NX + i]; }
                    PackedSink s{fin + L::NCFIN};
                    Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
                }
                // backward chunks: lane = stage evaluates F, G, Hxx, Hxu, Huu at (x_t, u_t, lambda_{t+1}); defect and Lagrangian gradients come from the residual set
                for (int g = 0; g < nchunk && !stop(); ++g) {
                    int t0, cnt;
                    bchunk(g, t0, cnt);
                    if (g >= 2) {
                        bool freed = false;
                        int n = 0;
                        while (!(freed = ms2_load(ctl + MS2_CONS) >= g - 1) && !stop()) { __builtin_amdgcn_s_sleep(2); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!freed) break;
                    }
                    if (lane < cnt) {
                        PDP_MS2_PAR();
                        const int t = t0 + lane;
                        double xc[NX], uc[NU], lc[NX];
                        double* row = pool + (g & 1) * L::BUF + lane * BS;
#pragma unroll
                        for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; lc[i] = lb[t * NX + i]; row[L::C0 + i] = rc[t * NX + i]; row[L::RX + i] = rx[t * NX + i]; }
#pragma unroll
                        for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; row[L::RU + i] = ru[t * NU + i]; }
                        PackedSink s{row};
                        Mdl::eval_sol(xc, uc, lc, th, pc, s);
                        row[L::CB0] = 0.0;
#pragma unroll
                        for (int i = 0; i < Mdl::SOL_NCONST; ++i) row[L::CB0 + 1 + i] = Mdl::sol_const(i);
                    }
                    f3_signal(ctl + MS2_PROD, g + 1);
                }
                // forward chunks: F', G', the defect and the gradients the directional derivative needs; behind each consumed chunk the multiplier step
                for (int c = 0; c < nchunkF && !stop(); ++c) {
                    const int g = nchunk + c, t0 = c * chF, cnt = min(chF, T - t0);
                    if (g >= 2) {
                        bool freed = false;
                        int n = 0;
                        while (!(freed = ms2_load(ctl + MS2_CONS) >= g - 1) && !stop()) { __builtin_amdgcn_s_sleep(2); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!freed) break;
                    }
                    if (lane < cnt) {
                        PDP_MS2_PAR();
                        const int t = t0 + lane;
                        double xc[NX], uc[NU];
                        double* row = pool + (g & 1) * L::BUF + lane * FS;
#pragma unroll
                        for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; row[L::FC0 + i] = rc[t * NX + i]; row[L::FRX + i] = rx[(t + 1) * NX + i]; }
#pragma unroll
                        for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; row[L::FRU + i] = ru[t * NU + i]; }
                        PackedSink s{row};
                        Mdl::eval_solf(xc, uc, nullptr, th, pc, s);
                        row[L::CF0] = 0.0;
#pragma unroll
                        for (int i = 0; i < Mdl::SOLF_NCONST; ++i) row[L::CF0 + 1 + i] = Mdl::solf_const(i);
                    }
                    f3_signal(ctl + MS2_PROD, g + 1);
                    if (c >= 1) {       // the runner has left chunk c - 1 (it could not start chunk c before the signal above): its dx are in memory
                        bool got = false;
                        int n = 0;
                        while (!(got = ms2_load(ctl + MS2_CONS) >= g) && !stop()) { __builtin_amdgcn_s_sleep(2); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!got) break;
                        dlam_chunk((c - 1) * chF, min(chF, T - (c - 1) * chF));
                    }
                }
                if (!stop()) {
                    bool got = false;
                    int n = 0;
                    while (!(got = ms2_load(ctl + MS2_CONS) >= nchunk + nchunkF) && !stop()) { __builtin_amdgcn_s_sleep(2); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                    if (got) dlam_chunk((nchunkF - 1) * chF, min(chF, T - (nchunkF - 1) * chF));
                }
                __threadfence_block();
            }
            if (dead) break;
            f3_signal(ctl + MS2_DONE, last);
        }
    }
#undef PDP_MS2_PAR
}

template <class Mdl>
__host__ inline int64_t ms2_ws_bytes(int B, int T, int max_iter) { return (int64_t)B * Ms2Layout<Mdl>::ws_doubles(T, max_iter < 0 ? 0 : max_iter) * (int64_t)sizeof(double); }

}  // namespace pdp
