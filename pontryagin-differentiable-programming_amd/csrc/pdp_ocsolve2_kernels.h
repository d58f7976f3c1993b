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
//                residual pass, no dyn / costate / H_u re-evaluation inside the sweep).
// The runner sends commands (sweep / trial / trial-then-sweep / exit) through a mailbox in LDS; chunks are handed over with produced / consumed
// counters (release / acquire at workgroup scope); every wait has a watchdog (a protocol error ends the trajectory with PDP_MS_INTERNAL
// instead of hanging the GPU).
//
// What else changed against the one-wave kernel (each step measured, profiles/r03_ms2_phase_timing_v*.txt, r03_ms2_variants.txt):
//   * homogeneous form of the Newton step for 4 < n < 16 (riccati_backward_aug): the affine column of the LQ problem becomes row / column n + 1 of
//     F~ = [F c; 0 1], Hxx~ = [Hxx rx; rx' 0], P~ = [P W; W' s] - 13 + 9 MFMAs per backward step instead of 18 + 10, and one record per stage;
//   * the multiplier step dlam_t' = x~_{t+1}' P~_{t+1} is four small MFMAs on the runner, off its dependency chain (x~ is carried in every tile column);
//     the evaluator's lane-per-stage version of it (dlam_staged / dlam_chunk below) remains for the n <= 4 and n = 16 forms;
//   * the iterate exists twice, stage-minor (Ms2Layout::group_doubles): a trial point is written into the other point set, accepting it is an index
//     flip - there is no update pass;
//   * the line search issues TRIAL_SWEEP: the evaluator continues with the first chunks of the next sweep at the trial point while the runner
//     decides; a rejected trial costs an abort of that sweep;
//   * the trial pass works with one NODE per lane and hands x_{t+1} - f(x_t, u_t) up with __shfl_up; stores are range-checked buffer stores.
//
// Placement.  TPW trajectories per workgroup of 2 TPW waves, runner = wave j, evaluator = wave j + TPW.  TPW = 4 (512 threads): the pair
// shares a SIMD (probes/wave_placement_probe.hip) - 1024 trajectories fill the chip with every SIMD running one runner and its
// evaluator.  TPW = 1 / 2 (small batches, e.g. C2's 256 problems or a 512-trajectory shard): the pair sits on two different SIMDs of
// the CU, nothing is shared but the LDS.
//
// Workspace traffic.  Per stage the sweep leaves K [m x n], k [m], W_{t+1} [n] and P_{t+1} - for n > 4 only its upper triangle (P is
// symmetrised every step, so the two halves are bit-identical); in the homogeneous form K~ = [K | k] (m x (n+1)) and the upper triangle of
// P~ ((n+1)(n+2)/2): 56 + 105 doubles for the quadrotor against 238.  The runner's forward steps re-read them, requested TWO steps ahead
// (three for n <= 4) through buffer loads (absent tile elements are out-of-range lanes: they load 0 and store nothing, no sink words, no
// predicated blocks).
#pragma once
#include <type_traits>
#include "pdp_ocsolve_kernels.h"
#include "pdp_fused3_kernels.h"

#ifndef PDP_MS2_DLAM_STAGED
#define PDP_MS2_DLAM_STAGED 1      // the multiplier step reads its (P, W) records through an LDS copy (dlam_staged) when four trajectories share a CU; 0: straight from the workspace
#endif
#ifndef PDP_MS2_SLEEP
#define PDP_MS2_SLEEP 2             // s_sleep argument of the hand-over polls (64 cycles each)
#endif
#ifndef PDP_MS2_TAILF
#define PDP_MS2_TAILF 0             // stages of a short last forward chunk (0: equal chunks - measured best, profiles/r03_ms2_variants.txt)
#endif

namespace pdp {

template <class Mdl>
struct Ms2Layout {
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    static constexpr bool SMALL = NX <= 4;
    // AUG: the Newton step in homogeneous form (riccati_backward_aug, pdp_riccati.h): state augmented by a constant 1, NA = NX + 1 <= 16 rows / columns
    static constexpr bool AUG = !SMALL && NX < 16;
    static constexpr int NA = AUG ? NX + 1 : NX;
    static constexpr int NS = Mdl::SOL_NVAR, NF = Mdl::SOLF_NVAR;
    // backward pool row: [sol entries | defect c_t (NX) | grad_x L (NX) | grad_u L (NU) | 0.0 | constants]  (uniform rows, see Fused3Layout)
    static constexpr int C0 = NS, RX = NS + NX, RU = NS + 2 * NX, CB0 = NS + 2 * NX + NU;
    static constexpr int ONEB = CB0 + 1 + Mdl::SOL_NCONST;                 // a 1.0 behind the constants (the homogeneous coordinate of F~)
    static constexpr int BSTRIDE = (ONEB + 1) | 1;
    // forward pool row: [solf entries | defect c_t (NX) | grad_x L of stage t+1 (NX) | grad_u L (NU) | 0.0 | constants]
    static constexpr int FC0 = NF, FRX = NF + NX, FRU = NF + 2 * NX, CF0 = NF + 2 * NX + NU;
    static constexpr int ONEF = CF0 + 1 + Mdl::SOLF_NCONST;
    static constexpr int FSTRIDE = (ONEF + 1) | 1;
    // a trajectory's LDS slice (doubles)
    static constexpr int FIN = RICCATI_SCRATCH;                          // [0.0 | terminal constants | terminal entries]
    static constexpr int NCFIN = 1 + Mdl::FIN_NCONST;
    static constexpr int PAR = FIN + NCFIN + Mdl::FIN_NVAR;               // theta (NP) | theta-only precomputed values (NPC)
    static constexpr int DLT = PAR + NP + Mdl::NPC;                       // NX: terminal gradient h_x(x_T) - lambda_T
    static constexpr int CTL = (DLT + NX + 1) & ~1;                       // mailbox: 16 ints | 32 doubles
    static constexpr int POOL = CTL + 40;                              // (mailbox: 16 ints | 32 doubles)
    static constexpr int SLICE = 160 * 1024 / 8 / 4;
    static constexpr int BUF = (SLICE - POOL) / 2;
    static constexpr int ROWS = BUF / BSTRIDE < 64 ? BUF / BSTRIDE : 64;
    static constexpr int ROWSF = BUF / FSTRIDE < 64 ? BUF / FSTRIDE : 64;
    // workspace per stage
    // gains: K [NU x NX] | k [NU]; homogeneous form: K~ = [K | k] [NU x NA]
    static constexpr int GSZ = NX * NU + NU;
    // P_{t+1}: small systems keep the full matrix (rep form) followed by W_{t+1}; else the upper triangle of P (followed by W), or - homogeneous form -
    // of P~ = [P W; W' s], which contains W as its last column
    static constexpr int PSZ = SMALL ? NX * NX : NA * (NA + 1) / 2;
    static constexpr int PWSZ = AUG ? PSZ : PSZ + NX;
    __host__ __device__ static constexpr int pk(int i, int j) { return SMALL ? i * NX + j : (i <= j ? i * NA - i * (i - 1) / 2 + (j - i) : j * NA - j * (j - 1) / 2 + (i - j)); }
    // offsets inside a stage's records: K[i][k], k_i (gains) and W_i (behind / inside P)
    __host__ __device__ static constexpr int gK(int i, int k) { return AUG ? i * NA + k : i * NX + k; }
    __host__ __device__ static constexpr int gk(int i) { return AUG ? i * NA + NX : NX * NU + i; }
    __host__ __device__ static constexpr int pW(int i) { return AUG ? pk(i, NX) : PSZ + i; }
    // Everything the evaluator touches with one lane per stage is kept STAGE-MINOR: element (stage t, component i) at [i * TS + t], TS = T + 1 - a wave
    // instruction of such a pass reads 64 consecutive doubles (4 cache lines).  In the API's stage-major layout [t][i] every lane sits in its own cache
    // line: at B = 1024 (four trajectories per CU) the trial pass alone kept a CU's address unit busy for ~40 k cycles (profiles/r03_ms2_phase_timing_v1.txt).
    // One group = [x-like (NX rows) | u-like (NU rows) | lambda-like (NX rows)] = (2 NX + NU) * TS doubles:
    //     point set 0 | point set 1 (the iterate is in one, the trial point goes to the other) | step (dx | du | dlam) |
    //     residual set 0 | residual set 1 (grad_x L | grad_u L | defect c) |
    //     second-order correction: its constraint block c_soc (the x-like rows) | the plain step, kept while corrected steps are tried
    // followed by the gains, (P, W) and the filter.  The API arrays are read once (warm start) and written once (the result).
    // PDP_MS_PREDICT inside the kernel: staging block of the sensitivity rows (64 rows of NP, or 64 / NX stages of the Riccati record) + dx of every node, in the pool
    // (third candidate: 64 / NX stages of the packed fp32 record X | U | tri(P) | W plus the block's dx - with many controls and few states, e.g. n = 2, m = 3, the U part
    // makes it the largest; a model of that shape did not compile before round 5)
    static constexpr int PRED_STG_A = 64 * NP > (64 / NX) * (NX * NX + NX * NP + 1) ? 64 * NP : (64 / NX) * (NX * NX + NX * NP + 1);
    static constexpr int PRED_STG_R = ((64 / NX) * (2 * NX * NP + NU * NP + NX * (NX + 1) / 2) + 1) / 2 + (64 / NX) * NX + 2;
    static constexpr int PRED_STG = PRED_STG_A > PRED_STG_R ? PRED_STG_A : PRED_STG_R;
    __host__ __device__ static constexpr bool predict_fits(int T) { return (int64_t)(T + 1) * NX + PRED_STG <= 2 * BUF; }
    __host__ __device__ static constexpr int64_t group_doubles(int T) { return (int64_t)(2 * NX + NU) * (T + 1); }
    __host__ __device__ static constexpr int64_t ws_doubles(int T, int max_iter) {
        return 7 * group_doubles(T) + (int64_t)T * GSZ + (int64_t)T * PWSZ + 2 * (int64_t)(max_iter + 1) + 2 * group_doubles(T);      // (the last two groups: the watchdog's stored iterate and direction)
    }
};

template <class Mdl>
__host__ __device__ constexpr bool ms2_ok() {
    using L = Ms2Layout<Mdl>;
    return Mdl::NX <= 16 && Mdl::NU <= 4 && L::ROWS >= 4 && L::ROWSF >= 4 && Mdl::FIN_NVAR + L::NCFIN + L::PAR <= L::SLICE;
}


// mailbox slots (ints) and result slots (doubles behind them)
enum { MS2_SEQ = 0, MS2_TYPE = 1, MS2_PROD = 2, MS2_CONS = 3, MS2_DONE = 4, MS2_ABORT = 5, MS2_DEAD = 6, MS2_CUR = 7, MS2_DST = 8, MS2_TDONE = 9,
       MS2_PDONE = 10,         // PDONE: the runner's (primal) half of a line-search trial is in memory
       MS2_CSRC = 11,          // SWEEP: 1 = the defect column of the LQ problem comes from the c_soc array (second-order correction) instead of the residual set
       MS2_SOCM = 12, MS2_SOCN = 13,        // the runner's own: a correction is under way / corrected points tried in this iteration (see the line search)
       MS2_GDONE = 14,         // the evaluator's pass over the previous solution (prediction guard), done while the runner forms the prediction, is in the mailbox
       MS2_PARS = 15 };        // theta and its precomputed values are in LDS (the evaluator's guard pass starts before the first command)
// SWEEP: chunks of the iterate in set CUR.  TRIAL: residuals of CUR + alpha step -> set DST.  TRIAL_SWEEP: the same trial, then - speculating that the
// runner accepts the point - straight on with the sweep of set DST (the runner aborts it otherwise)
// RESTORE: the states of set CUR replaced by the rollout of its controls, its multipliers by zero, then the residuals of that point (as TRIAL with alpha = 0)
enum { MS2_CMD_EXIT = 0, MS2_CMD_SWEEP = 1, MS2_CMD_TRIAL = 2, MS2_CMD_TRIAL_SWEEP = 3, MS2_CMD_RESTORE = 4 };
enum { MS2_ALPHA = 0, MS2_F = 1, MS2_TH = 2, MS2_PR = 3, MS2_DU = 4, MS2_Z = 5, MS2_L = 6, MS2_LC = 7, MS2_FIN = 8,
       MS2_S_GD = 9, MS2_S_AMIN = 10, MS2_S_THOLD = 11,        // the line search's state while the sweep of a correction runs (slots 12 .. 19: timing builds)
       MS2_G0 = 24 };          // slots 24 .. 31: the guard pass's copy of MS2_F .. MS2_FIN

// Mailbox values come out of LDS in vector registers although every lane reads the same word: said explicitly (v_readfirstlane), or every pointer and
// branch derived from them would be treated as divergent - 64-bit per-lane addresses for each of the trial pass's ~100 loads, masked branches in the
// runner's control flow
PDP_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
PDP_DEV double uni(double v) { return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v))); }
// A value the optimiser cannot see through: per-lane maps derived from opaque(lane) INSIDE a sweep are recomputed at every sweep (a few hundred cycles against
// the sweep's 50 - 100 k) instead of being hoisted out of the iteration loop, where the maps of BOTH sweeps stayed live across each other and pushed the
// four-trajectories-per-workgroup instantiation (256 registers per wave) into scratch memory (round 3: 21 spilled VGPRs)
PDP_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
// the same for values that live in SCALAR registers (launch constants: T, the workspace pointers).  Row offsets and row pointers derived from the plain T + 1 or from `stp`
// are invariants of the whole launch: the compiler forms all of them at kernel entry - two scalar registers per row pointer, 2 NX + NU rows per array - and keeps them, i.e.
// parks them in lanes of vector registers (round 6: FOUR vector registers of the four-trajectory instantiation held ~200 such words, read back ~1000 times).  Derived from an
// opaque copy inside a pass they are formed there (a few scalar instructions beside thousands of vector ones) and die with it.
PDP_DEV int sopaque(int v) { asm volatile("" : "+s"(v)); return v; }
template <class P> PDP_DEV P* sopaque(P* p) { asm volatile("" : "+s"(p)); return p; }
PDP_DEV int ms2_load(int* f) { return uni(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); }
// wait until *f >= v; false when the partner never gets there (watchdog) or the trajectory has been declared dead
PDP_DEV bool ms2_wait_ge(int* f, int v, int* ctl) {
    int n = 0;
    while (ms2_load(f) < v) {
        __builtin_amdgcn_s_sleep(PDP_MS2_SLEEP);
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

// Cross-lane traffic of this kernel through DPP modifiers instead of ds_bpermute (round 6): the shuffle form keeps one address register per distance - (lane ^ o) << 2, the
// same values at every call site, so the compiler shares them across the WHOLE kernel: six registers live through both sweeps of an instantiation that sits on its 256 -
// and sends every word through the LDS crossbar.  ms2_sum / ms2_max: rotations inside the rows of 16 lanes (row_ror 1, 2, 4, 8: every lane of a row holds the row's
// total), then row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3; lane 63 holds the total, returned through v_readlane (uniform).  The ORDER of the
// additions differs from wave_sum's butterfly: sums agree to rounding, not bit for bit.
template <int CTRL, int ROWMASK = 0xf>
PDP_DEV double ms2_dpp(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
PDP_DEV double ms2_sum(double v) {
    v += ms2_dpp<0x121>(0.0, v); v += ms2_dpp<0x122>(0.0, v); v += ms2_dpp<0x124>(0.0, v); v += ms2_dpp<0x128>(0.0, v);
    v += ms2_dpp<0x142, 0xa>(0.0, v);
    v += ms2_dpp<0x143, 0xc>(0.0, v);
    return readlane_f64(v, 63);
}
PDP_DEV double ms2_max(double v) {
    v = fmax(v, ms2_dpp<0x121>(v, v)); v = fmax(v, ms2_dpp<0x122>(v, v)); v = fmax(v, ms2_dpp<0x124>(v, v)); v = fmax(v, ms2_dpp<0x128>(v, v));
    v = fmax(v, ms2_dpp<0x142, 0xa>(v, v));
    v = fmax(v, ms2_dpp<0x143, 0xc>(v, v));
    return readlane_f64(v, 63);
}
// lane l <- lane l - 1 (wave_shr:1); lane 0 keeps its own value, like __shfl_up(v, 1)
PDP_DEV double ms2_up1(double v) { return ms2_dpp<0x138>(v, v); }

// stage-minor access: UNIFORM row pointer (array base + component * stride: scalar registers) + this lane's byte offset (8 * stage, one VGPR shared by
// every access of the pass) - the saddr + voffset form of global_load / global_store; `row[i * TS + t]` would carry a 64-bit address per component
PDP_DEV double sm_ld(const double* row, unsigned off8) { return *(const double*)((const char*)row + off8); }
PDP_DEV void sm_st(double* row, unsigned off8, double v) { *(double*)((char*)row + off8) = v; }

// WD: IPOPT's watchdog in the line search (PDP_MS_WITH_WATCHDOG; see the main loop).  A template constant, so that the instantiations without it are the code they were:
// the four-trajectory instantiation sits on its 256 registers, and the watchdog is only instantiated for one / two trajectories per workgroup (512 registers per wave).
template <class Mdl, int TPW, bool WD = false>
__global__ void __launch_bounds__(128 * TPW) oc_solve_ms2_kernel(int B, int T, pdp_oc_ms_opts op, const double* __restrict__ x0, const double* __restrict__ theta,
                                                                  int tb, double* __restrict__ x, double* __restrict__ u, double* __restrict__ lam,
                                                                  double* __restrict__ cost, double* __restrict__ resid, int32_t* __restrict__ converged,
                                                                  int32_t* __restrict__ iters, int32_t* __restrict__ status, double* __restrict__ gains_out,
                                                                  double* __restrict__ iter_log, double* __restrict__ ws) {
    using L = Ms2Layout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, M = NU;
    constexpr bool SMALL = L::SMALL, AUG = L::AUG;
    constexpr int NRT = SMALL ? 1 : 4, NA = L::NA;
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
    // theta and the theta-only precomputed values are read from LDS where the generated code uses them (broadcast reads): copied into registers in front of every pass they
    // were 2 (NP + NPC) vector registers holding uniform values across the whole pass - 64 for the quadrotor, 52 for the rocket, in passes that sit on the register limit
#define PDP_MS2_PAR()                                                     \
    const double* th = par;                                               \
    const double* pc = par + NP
    double* xb = x + (int64_t)b * (T + 1) * NX;            // API arrays (stage-major): read at the start (warm), written at the end
    double* ub = u + (int64_t)b * T * NU;
    double* lb = lam + (int64_t)b * T * NX;
    double* w0 = ws + (int64_t)b * L::ws_doubles(T, op.max_iter);
    const int TS = T + 1;                                    // stride of the stage-minor arrays
    const int64_t GRP = L::group_doubles(T);
    const int OU = NX * TS, OL = (NX + NU) * TS;             // u-like / lambda-like rows inside a group
    auto Pt = [&](int k) { return w0 + k * GRP; };           // point sets 0 / 1: x (t, i) at [i TS + t], u at [OU + i TS + t], lambda at [OL + i TS + t]
    double* stp = w0 + 2 * GRP;                              // step: dx | du | dlam, same addressing
    auto Rs = [&](int k) { return w0 + (3 + k) * GRP; };     // residual sets 0 / 1: grad_x L (T + 1 stages) | grad_u L | defect c
    double* csoc = w0 + 5 * GRP;                             // second-order correction: c_soc (t, i) at [i TS + t]
    double* stpb = w0 + 6 * GRP;                             //                          the plain step while corrected ones are tried
    double* gw = w0 + 7 * GRP;                               // gains, T x GSZ
    double* pw = gw + (int64_t)T * GSZ;                      // P_{t+1}, W_{t+1}, T x PWSZ
    double* fth = pw + (int64_t)T * PWSZ;                    // filter: theta entries (at most one per iteration) ...
    double* fph = fth + (op.max_iter + 1);                   //         ... and phi entries
    double* wdp = fph + (op.max_iter + 1);                   // watchdog: the stored iterate ...
    double* wds = wdp + GRP;                                 //           ... and its direction
    // chunks: backward chunk g (0 = last stages) covers [t0, t0 + cnt); forward chunks follow in the same numbering
    const int nchunk = (T + L::ROWS - 1) / L::ROWS;
    const int ch = (T + nchunk - 1) / nchunk;
    auto bchunk = [&](int g, int& t0, int& cnt) { const int c = nchunk - 1 - g; t0 = c * ch; cnt = min(ch, T - t0); };
    // forward chunks: equal chunks (optionally, PDP_MS2_TAILF > 0, a short last chunk behind equal chunks over the first T - tailF stages).  The multiplier step
    // of a chunk runs on the evaluator behind the runner - hidden for every chunk but the last, whose (P, W) records all trajectories of the batch fetch at the
    // same moment (21 MB in one burst for 25 stages at B = 1024: ~15 k cycles of memory time after the runner has finished).  A short last chunk was tried to
    // shrink that burst: the evaluator then still owes the multiplier step of the long chunk before it when the runner is done - no gain (profiles/r03_ms2_variants.txt)
    const int tailF = (PDP_MS2_TAILF > 0 && T > 3 * PDP_MS2_TAILF) ? PDP_MS2_TAILF : 0;
    const int nbodyF = (T - tailF + L::ROWSF - 1) / L::ROWSF, chF = (T - tailF + nbodyF - 1) / nbodyF;
    const int nchunkF = nbodyF + (tailF ? 1 : 0);
    auto fchunk = [&](int c, int& t0, int& cnt) { if (c < nbodyF) { t0 = c * chF; cnt = min(chF, T - tailF - t0); } else { t0 = T - tailF; cnt = tailF; } };

    // ---- the trial pass (both waves use it).  Residuals of the point (x, u, lambda) + a (dx, du, dlam), lane = NODE t = 0 .. T (T + 1 of them): node t < T
    // evaluates stage t at its own (x_t, u_t, lambda_{t+1}); what couples neighbouring stages - the defect c_{t-1} = f(x_{t-1}, u_{t-1}) - x_t and the
    // stationarity row grad_x L_t = H_x(t) - lambda_t - is formed by node t from ITS x_t and the previous node's f and lambda, handed up one lane (node 0 of a
    // later pass: from lane 63 of the pass before, through scalar registers); node T does the terminal terms.  No lane reads another stage's rows:
    // 2 (2 NX + NU) coalesced loads per pass, and every store is a range-checked buffer store (inactive lanes: out-of-range offset), so the pass has no
    // conditional block around memory operations.  a = 0: the current point (dx, du, dlam are not read into the result).
    // PART 0: everything (TRIAL / RESTORE commands, on the evaluator).  The line search splits the pass between the two waves - the runner would only wait:
    // PART 1, runner: the PRIMAL half - trial (x, u) into set `dst`, defects, objective, theta, the convergence measures of the primal side; it is all the
    //         filter needs, so the runner decides on the trial point without waiting for anybody;
    // PART 2, evaluator: the DUAL half - trial lambda, grad_x L, grad_u L and their measures (used once the point is accepted), then on with the sweep.
    double a_f = 0.0, a_th = 0.0, a_pr = 0.0, a_du = 0.0, a_z = 0.0, a_l = 0.0, a_lc = 0.0;
    bool fin_all = true;
    // PART 3, runner: everything, like PART 0, but the sums stay in this wave's registers (a_f .. a_lc, fin_all) and the mailbox's result slots are left alone - the
    //         prediction guard's pass over the previous solution, which runs BESIDE the evaluator's pass over the predicted point (see the guard).
    auto trial_pass = [&](auto part_tag, double a, int cur, int dst) {
        constexpr int PART = decltype(part_tag)::value;
        constexpr bool PRIMAL = PART != 2, DUAL = PART != 1, KEEP = PART == 3;
        PDP_MS2_PAR();
        const int TSl = sopaque(TS), OUl = NX * TSl, OLl = (NX + NU) * TSl;      // (see sopaque)
        const double* stpl = sopaque((const double*)stp);
        const bool stepped = a != 0.0, put = dst != cur;
        const double* __restrict__ ps = Pt(cur);
        const auto rsPd = __builtin_amdgcn_make_buffer_rsrc((void*)Pt(dst), 0, (int)(GRP * 8), 0x00020000);
        const auto rsRd = __builtin_amdgcn_make_buffer_rsrc((void*)Rs(dst), 0, (int)(GRP * 8), 0x00020000);      // grad_x L (T + 1 nodes; node 0: x_0 is fixed) | grad_u L | c
        auto bst = [](auto rs, unsigned soff, unsigned voff, double v_) {
            pdp_u2 w;
            w.x = (unsigned)__double2loint(v_); w.y = (unsigned)__double2hiint(v_);
            __builtin_amdgcn_raw_buffer_store_b64(w, rs, voff, soff, 0);
        };
        a_f = 0.0; a_th = 0.0; a_pr = 0.0; a_du = 0.0; a_z = 0.0; a_l = 0.0; a_lc = 0.0;
        bool fin = true;
        // Groups of 64 lanes OVERLAP by one node (round 6): lane 0 of a later group evaluates node `base` once more - it was lane 63 of the group before - only to hand its f and
        // lambda up to lane 1; it stores and sums nothing.  Carried from group to group instead (2 NX doubles held across the whole body of the pass, read with v_readlane)
        // they cost the pass registers it does not have, for a case - more than 64 nodes - that most horizons never meet.
        for (int base = 0; base == 0 || base < T; base += 63) {
            const int t = base + lane;
            const bool mine = !(base > 0 && lane == 0);
            const bool node = t <= T && mine, stage = t < T && mine, last = t == T && mine;
            const unsigned o8 = 8u * (unsigned)(t <= T ? t : T);      // (lanes behind the horizon read node T's slots and store nothing)
            const unsigned on = node ? o8 : MS2_OOB, os = stage ? o8 : MS2_OOB, oc = (node && t > 0) ? o8 - 8u : MS2_OOB;
            double xc[NX], uc[NU], lc[NX], v[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const double xa = sm_ld(ps + i * TSl, o8), xd = sm_ld(stpl + i * TSl, o8), la = sm_ld(ps + OLl + i * TSl, o8), ld = sm_ld(stpl + OLl + i * TSl, o8);
                xc[i] = stepped ? fma(a, xd, xa) : xa;
                lc[i] = stepped ? fma(a, ld, la) : la;                 // (slot T of the lambda / u rows exists and is unused: node T reads it and drops it)
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) { const double ua = sm_ld(ps + OUl + i * TSl, o8), ud = sm_ld(stpl + OUl + i * TSl, o8); uc[i] = stepped ? fma(a, ud, ua) : ua; }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if constexpr (PRIMAL) { a_z = fmax(a_z, node ? fabs(xc[i]) : 0.0); if (put) bst(rsPd, (unsigned)(i * TSl) * 8u, on, xc[i]); }
                if constexpr (DUAL) { a_l = fmax(a_l, stage ? fabs(lc[i]) : 0.0); if (put) bst(rsPd, (unsigned)(OLl + i * TSl) * 8u, os, lc[i]); }
            }
            if constexpr (PRIMAL) {
#pragma unroll
                for (int i = 0; i < NU; ++i) { a_z = fmax(a_z, stage ? fabs(uc[i]) : 0.0); if (put) bst(rsPd, (unsigned)(OUl + i * TSl) * 8u, os, uc[i]); }
            }
            double nl[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) {                              // the previous node's lambda
                nl[i] = ms2_up1(lc[i]);                                    // (lane 0: its own - node 0 has no predecessor, a later group's lane 0 uses none)
            }
            if constexpr (PRIMAL) {
                Mdl::dyn(xc, uc, th, pc, v);
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double nv = ms2_up1(v[i]);                     // the previous node's f(x, u)
                    const double ci = nv - xc[i];                        // defect of stage t - 1
                    bst(rsRd, (unsigned)(OLl + i * TSl) * 8u, oc, ci);
                    const bool has = node && t > 0;
                    a_th += has ? fabs(ci) : 0.0; a_pr = fmax(a_pr, has ? fabs(ci) : 0.0); a_lc += has ? nl[i] * ci : 0.0;
                    fin = fin && (!has || fabs(ci) <= 1.7e308);
                }
                double fT = 0.0;
                if (last) fT = Mdl::final_cost(xc, th, pc);              // (arithmetic only inside the branch)
                const double pcst = Mdl::path_cost(xc, uc, th, pc);
                a_f += stage ? pcst : (last ? fT : 0.0);
            }
            if constexpr (DUAL) {
                Mdl::costate_step(xc, uc, lc, th, pc, v);
                double hT[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) hT[i] = 0.0;
                if (last) Mdl::dhx(xc, th, pc, hT);                      // terminal node: h_x(x_T)
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const double g = last ? hT[i] - nl[i] : ((stage && t > 0) ? v[i] - nl[i] : 0.0);      // grad_x L of node t (x_0 is fixed: no row)
                    bst(rsRd, (unsigned)(i * TSl) * 8u, on, g);
                    a_du = fmax(a_du, node ? fabs(g) : 0.0);
                    fin = fin && (!node || fabs(g) <= 1.7e308);
                }
                double hu[NU];
                Mdl::dHu(xc, uc, lc, th, pc, hu);
#pragma unroll
                for (int i = 0; i < NU; ++i) { bst(rsRd, (unsigned)(OUl + i * TSl) * 8u, os, hu[i]); a_du = fmax(a_du, stage ? fabs(hu[i]) : 0.0); fin = fin && (!stage || fabs(hu[i]) <= 1.7e308); }
            }
        }
        if constexpr (PRIMAL) { a_f = ms2_sum(a_f); a_th = ms2_sum(a_th); a_lc = ms2_sum(a_lc); a_pr = ms2_max(a_pr); a_z = ms2_max(a_z); }
        if constexpr (DUAL) { a_du = ms2_max(a_du); a_l = ms2_max(a_l); }
        fin_all = __all(fin);
        if constexpr (DUAL && !KEEP) {                               // (the runner keeps its half - PART 3: all of it - in registers)
            if (lane == 0) {
                if constexpr (PRIMAL) { res[MS2_F] = a_f; res[MS2_TH] = a_th; res[MS2_PR] = a_pr; res[MS2_Z] = a_z; res[MS2_LC] = a_lc; }
                res[MS2_DU] = a_du; res[MS2_L] = a_l;
                res[MS2_FIN] = fin_all ? 1.0 : 0.0;
            }
        }
    };
    // The split pays where the two halves are comparable and the runner has registers to spare - the small systems (cart-pole cold solve 4.18 -> 3.94 ms).  For the
    // tile forms it does not: the quadrotor's trial pass is mostly `dyn`, and inside the runner's register allocation (256 VGPRs, gather maps of both sweeps live)
    // that half alone took longer than the whole pass on the evaluator (line search 23 k -> 30 k cycles, spills into the sweeps: 0.318 -> 0.340 ms; profiles/r03_ms2_variants.txt)
#ifndef PDP_MS2_SPLIT_ALL
#define PDP_MS2_SPLIT_ALL 0             // 1: the tile forms (n > 4) share the line search's trial pass between the two waves as well - measured slower twice (profiles/r06_ms2_variants.txt)
#endif
    constexpr bool SPLIT = SMALL || PDP_MS2_SPLIT_ALL;      // the line search's trial pass shared between the two waves
    constexpr bool SPLIT0 = SMALL;                           // ... and no speculative first sweep at the starting point (small systems)
    using PartAll = std::integral_constant<int, 0>;
    using PartPrimal = std::integral_constant<int, 1>;
    using PartDual = std::integral_constant<int, 2>;
    using PartKeep = std::integral_constant<int, 3>;

    // Stage-major API array [t][NC] <-> stage-minor workspace rows [i TS + t], by ONE wave, transposed through its LDS scratch so that BOTH sides are coalesced: lane = stage on
    // the workspace side (64 consecutive doubles per row and instruction), lane + 64 k on the API side.  Round 6 (probes/ms_timeline.py): written as `for (q = lane; ...)
    // dst[..] = src[q]` loops these copies were 24 dependent round trips to memory with one cache line per LANE on the workspace side - 24 k cycles for the starting point,
    // 17 k for the result, of a 500 k-cycle solve; with the loads batched but still scattered they slowed a trial pass running beside them by a third (the address unit).
    // (Index arithmetic from an opaque lane id, like the sweeps' maps: shared between the copies at the start and at the end of the launch it would stay live across the whole
    // iteration loop.)  `from_t`: first stage written (1: node 0 of the states is x_0, written by the caller).  (Tried and dropped: the runner writing the TRIAL point into the
    // API arrays beside the evaluator's trial pass, so that a solve ending with that trial has nothing left to write - the copy costs the pass what it saves at the end.)
    // rows_get: the API rows of the stages [0, ng), ng <= 64, of `src` (contiguous, NC per stage) into registers, lane = stage
    auto rows_get = [&](const double* __restrict__ src, int ng, auto nc_tag, auto& v, double* buf) {
        constexpr int NC = decltype(nc_tag)::value;
        constexpr int TB = RICCATI_SCRATCH / NC < 64 ? RICCATI_SCRATCH / NC : 64;      // stages per LDS batch
        constexpr int KQ = (TB * NC + 63) / 64;
        const int ln = opaque(lane);
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] = 0.0;
        for (int b0 = 0; b0 < ng; b0 += TB) {
            const int nq = min(TB, ng - b0) * NC;
            const double* s_ = src + (int64_t)b0 * NC;
            double w[KQ];
#pragma unroll
            for (int k = 0; k < KQ; ++k) { const int idx = ln + 64 * k; w[k] = s_[idx < nq ? idx : 0]; }
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < KQ; ++k) { const int idx = ln + 64 * k; if (idx < nq) buf[idx] = w[k]; }
            wave_lds_sync();
            if (ln >= b0 && ln < b0 + TB) {
#pragma unroll
                for (int i = 0; i < NC; ++i) v[i] = buf[(ln - b0) * NC + i];
            }
        }
        wave_lds_sync();
    };
    auto rows_in = [&](const double* __restrict__ src, int nt, auto nc_tag, double* dst, int from_t, double* buf) {
        constexpr int NC = decltype(nc_tag)::value;
        for (int g0 = 0; g0 < nt; g0 += 64) {
            double v[NC];
            rows_get(src + (int64_t)g0 * NC, min(64, nt - g0), nc_tag, v, buf);
            const int t = g0 + opaque(lane);
            if (t < nt && t >= from_t) {
#pragma unroll
                for (int i = 0; i < NC; ++i) dst[i * TS + t] = v[i];
            }
        }
    };
    auto rows_out = [&](const double* src, int nt, auto nc_tag, double* __restrict__ dst, double* buf) {
        constexpr int NC = decltype(nc_tag)::value;
        constexpr int TB = RICCATI_SCRATCH / NC < 64 ? RICCATI_SCRATCH / NC : 64;
        const int ln = opaque(lane);
        for (int g0 = 0; g0 < nt; g0 += 64) {
            const int t = g0 + ln < nt ? g0 + ln : nt - 1;
            double v[NC];
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] = src[i * TS + t];
            for (int b0 = 0; b0 < 64 && g0 + b0 < nt; b0 += TB) {
                const int nq = min(TB, nt - g0 - b0) * NC;
                wave_lds_sync();
                if (ln >= b0 && ln < b0 + TB) {
#pragma unroll
                    for (int i = 0; i < NC; ++i) buf[(ln - b0) * NC + i] = v[i];
                }
                wave_lds_sync();
                double* d_ = dst + (int64_t)(g0 + b0) * NC;
                for (int q = ln; q < nq; q += 64) d_[q] = buf[q];
            }
        }
        wave_lds_sync();
    };
    using TagNX = std::integral_constant<int, NX>;
    using TagNU = std::integral_constant<int, NU>;

    if (runner) {
        // ================================================ runner ================================================
        __builtin_amdgcn_s_setprio(3);
#ifdef PDP_MS_TIMING      // whole-launch timeline of the runner (row log_rows / 2 - 1 of the log): entry | starting point loaded | first residuals read | loop left | results written
        const long long tk0 = __builtin_readcyclecounter();
        long long tkw = 0, tkc = 0, tkp = 0;            // record prologue: cycles waiting for the loads of a batch | computing | before the first batch
#endif
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
            f3_signal(ctl + MS2_PARS, 1);
        }
        // ---- starting point: the caller's (x, u, lambda) [PDP_MS_WARM], or IPOPT's: w0 = 0 (PDP.py:155,166), x_0 = ini_state
        const bool warm = (op.flags & PDP_MS_WARM) != 0;
        // PDP_MS_PREDICT: the caller's point is the solution at the PREVIOUS parameter; what is loaded is its first-order prediction for the step dtheta,
        //     x_t + X_t dtheta,   u_t + U_t dtheta,   lam_t + P_{t+1} (X_{t+1} dtheta) + W_{t+1} dtheta
        // (X, U and the Riccati record of the gradient unit at that solution: pdp_oc_pdp_grad_sens_batched; the same numbers as pdp_oc_predict_batched, applied
        // here so that an IRL iteration needs neither another launch nor a copy of the trajectory).  dx is parked in the still unused pool for the multiplier part.
        const float* rec = (warm && (op.flags & PDP_MS_PREDICT) != 0 && op.dtheta) ? op.predict_record : nullptr;      // the packed fp32 record (PredRec) takes precedence
        const bool recp = rec && (op.flags & PDP_MS_PREDICT_PRIMAL) != 0;
        const bool pred = !rec && warm && (op.flags & PDP_MS_PREDICT) != 0 && op.dtheta && op.dxdp && op.dudp;
        const bool predl = pred && op.riccati != nullptr;
        double corr_l = 0.0;           // largest predicted change of a state or control against max(1, |its value|): the guard below trusts small corrections unseen
        // (only the comparison with PDP_MS_GUARD_TRUST is ever used: |d| <= trust max(1, |v|) says the same without the division - fifty of them per lane since the prediction runs
        // with one lane per stage; a non-finite correction fails the comparison and is never trusted)
        auto corr_upd = [&](double d, double v) { corr_l = fabs(d) <= PDP_MS_GUARD_TRUST * fmax(1.0, fabs(v)) ? corr_l : 1e300; };
        {
            double* s0 = Pt(0);
            constexpr int RSZ = oc_riccati_doubles<Mdl>();
            if (rec) {
                // Round 6.  The records of a BATCH of stages (as many as the pool holds - all of them for the X | U part of a quadrotor record at T = 50; contiguous in
                // memory) go from memory straight into LDS (global_load_lds: no registers, a kilobyte per instruction for whole records), every lane's x_{t+1}, lambda_t,
                // u_t of the batch are requested behind them, and ONE wait covers all of it.  Then lane = item (row i, stage sg) WITH THE STAGE RUNNING FASTEST - the stores
                // into the stage-minor rows are runs of consecutive doubles - forms dx_{t+1}, du_t, parks dx in LDS, and, after the exchange,
                // dlam_t = W_{t+1} dtheta + P_{t+1} dx_{t+1}.  Until round 6: blocks of 64 / NX stages, three dependent round trips to memory per block (records -> registers
                // -> LDS, then x / u, then lambda), the row running fastest (every lane of a store in its own cache line): 70 k cycles of a 510 k-cycle solve at C3
                // (probes/ms_timeline.py), the runner's alone.  (Also measured: one lane per stage with literal offsets - no index arithmetic, no exchange, but a quarter of the
                // lanes busy: twice as slow.)  Same sums in the same order as before (and as oc_predict_kernel).
                using R = PredRec<Mdl>;
                constexpr int RD = NX, RDU = NU;                    // rounds of 64 items a batch of 64 stages needs
                constexpr int QP = (R::P + 3) / 4;                  // 16-byte words that cover the X | U part of a record
                const int RS = recp ? 4 * QP : R::SIZE;             // floats per stage in LDS (PDP_MS_PREDICT_PRIMAL: the X | U part of every record only, rounded up to whole 16-byte words)
                const int chcap_ = (4 * L::BUF - 320) / (RS + (recp ? 0 : 2 * NX));      // (256 floats of slack behind the records: the last load instruction of a batch writes a whole kilobyte)
                const int chcap = chcap_ < 64 ? chcap_ : 64;
                const int nbat = (T + chcap - 1) / chcap, CH = (T + nbat - 1) / nbat;
                float* stage = (float*)pool;
                double* dxb = pool + (chcap * RS + 256 + 1) / 2;   // dx of the batch: [stage][row]
                double dth[NP > 0 ? NP : 1];
#pragma unroll
                for (int j = 0; j < NP; ++j) dth[j] = op.dtheta[(int64_t)b * op.dtheta_bstride + j];
                for (int i = lane; i < NX; i += 64) s0[i * TS] = x0[(int64_t)b * NX + i];
#ifdef PDP_MS_TIMING
                tkp = __builtin_readcyclecounter() - tk0;
#endif
                for (int t0 = 0; t0 < T; t0 += CH) {
#ifdef PDP_MS_TIMING
                    const long long tb0_ = __builtin_readcyclecounter();
#endif
                    const int nst = min(CH, T - t0), nd = nst * RS;
                    const float* s_ = rec + ((int64_t)b * T + t0) * R::SIZE;
                    if (recp) {                                     // 16 bytes per lane here too: lane -> (stage, word) - a wave has 63 loads in flight at most, and as 120
                        for (int k0 = 0; k0 < nst * QP; k0 += 64) {  // dword loads per batch the X | U parts came in at 3 TB/s, chip-wide
                            int iq = k0 + lane;
                            iq = iq < nst * QP ? iq : 0;
                            const int si = iq / QP;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + si * R::SIZE + 4 * (iq - si * QP)),
                                                             (__attribute__((address_space(3))) void*)(stage + 4 * k0), 16, 0, 0);      // (the last word of a stage may reach into its P part: inside the record)
                        }
                    } else {                                        // whole records: 16 bytes per lane and instruction (global_load_lds_dwordx4, gfx950)
                        for (int k0 = 0; k0 < nd; k0 += 256) {
                            int idx = k0 + 4 * lane;
                            idx = idx + 4 <= nd ? idx : 0;         // (no read beyond the batch; the words behind the last whole quadruple come in with the dword form below)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + idx), (__attribute__((address_space(3))) void*)(stage + k0), 16, 0, 0);
                        }
                        const int tail = nd & ~3;
                        if (tail < nd)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s_ + (tail + lane < nd ? tail + lane : tail)), (__attribute__((address_space(3))) void*)(stage + tail), 4, 0, 0);      // (behind the quadruples: loads land in order; lanes beyond nd write into the slack)
                    }
                    const int mdiv = ((1 << 20) + nst - 1) / nst;   // it_ / nst = (it_ * mdiv) >> 20, exact for it_ < 1024
                    const int nrx = (nst * NX + 63) >> 6, nru = (nst * NU + 63) >> 6;      // rounds of this batch
                    double xv[RD], lv[RD], uv[RDU];
#pragma unroll
                    for (int rd = 0; rd < RD; ++rd) {
                        xv[rd] = 0.0; lv[rd] = 0.0;
                        if (rd < nrx) {
                            const int it_ = lane + 64 * rd, ok_ = it_ < nst * NX, i = ok_ ? (it_ * mdiv) >> 20 : 0, sg = ok_ ? it_ - i * nst : 0, q = (t0 + sg) * NX + i;
                            xv[rd] = xb[NX + q];
                            lv[rd] = lb[q];
                        }
                    }
#pragma unroll
                    for (int rd = 0; rd < RDU; ++rd) {
                        uv[rd] = 0.0;
                        if (rd < nru) {
                            const int it_ = lane + 64 * rd, ok_ = it_ < nst * NU, iu = ok_ ? (it_ * mdiv) >> 20 : 0, sg = ok_ ? it_ - iu * nst : 0;
                            uv[rd] = ub[(t0 + sg) * NU + iu];
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef PDP_MS_TIMING
                    const long long tb1_ = __builtin_readcyclecounter();
                    tkw += tb1_ - tb0_;
#endif
#pragma unroll
                    for (int rd = 0; rd < RD; ++rd) {
                        const int it_ = lane + 64 * rd;
                        if (rd < nrx && it_ < nst * NX) {
                            const int i = (it_ * mdiv) >> 20, sg = it_ - i * nst, t = t0 + sg;
                            const float* r = stage + sg * RS;
                            double dx = 0.0;
#pragma unroll
                            for (int j = 0; j < NP; ++j) dx = fma((double)r[R::X + i * NP + j], dth[j], dx);
                            if (!recp) dxb[sg * NX + i] = dx;
                            corr_upd(dx, xv[rd]);
                            s0[i * TS + t + 1] = xv[rd] + dx;
                            if (recp) s0[OL + i * TS + t] = lv[rd];
                        }
                    }
#pragma unroll
                    for (int rd = 0; rd < RDU; ++rd) {
                        const int it_ = lane + 64 * rd;
                        if (rd < nru && it_ < nst * NU) {
                            const int iu = (it_ * mdiv) >> 20, sg = it_ - iu * nst, t = t0 + sg;
                            const float* r = stage + sg * RS;
                            double du = 0.0;
#pragma unroll
                            for (int j = 0; j < NP; ++j) du = fma((double)r[R::U + iu * NP + j], dth[j], du);
                            corr_upd(du, uv[rd]);
                            s0[OU + iu * TS + t] = uv[rd] + du;
                        }
                    }
                    wave_lds_sync();
                    if (!recp) {
#pragma unroll
                        for (int rd = 0; rd < RD; ++rd) {
                            const int it_ = lane + 64 * rd;
                            if (rd < nrx && it_ < nst * NX) {
                                const int i = (it_ * mdiv) >> 20, sg = it_ - i * nst, t = t0 + sg;
                                const float* r = stage + sg * RS;
                                double dl = 0.0;
#pragma unroll
                                for (int j = 0; j < NP; ++j) dl = fma((double)r[R::W + i * NP + j], dth[j], dl);
#pragma unroll
                                for (int k = 0; k < NX; ++k) dl = fma((double)r[R::P + R::tri(i, k)], dxb[sg * NX + k], dl);
                                s0[OL + i * TS + t] = lv[rd] + dl;
                            }
                        }
                    }
                    wave_lds_sync();
#ifdef PDP_MS_TIMING
                    tkc += __builtin_readcyclecounter() - tb1_;
#endif
                }
            } else if (!pred) {
                if (warm) {
#ifdef PDP_MS_TIMING
                    tkp = __builtin_readcyclecounter() - tk0;
#endif
                    for (int i = lane; i < NX; i += 64) s0[i * TS] = x0[(int64_t)b * NX + i];
                    rows_in(xb, T + 1, TagNX{}, s0, 1, scratch);
                    rows_in(ub, T, TagNU{}, s0 + OU, 0, scratch);
                    rows_in(lb, T, TagNX{}, s0 + OL, 0, scratch);
                } else {
                for (int q = lane; q < (T + 1) * NX; q += 64) { const int t = q / NX, i = q - t * NX; s0[i * TS + t] = t == 0 ? x0[(int64_t)b * NX + i] : 0.0; }
                for (int q = lane; q < T * NU; q += 64) { const int t = q / NU, i = q - t * NU; s0[OU + i * TS + t] = 0.0; }
                for (int q = lane; q < T * NX; q += 64) { const int t = q / NX, i = q - t * NX; s0[OL + i * TS + t] = 0.0; }
                }
            } else {
                // The sensitivity arrays are row-major [row][NP] (a row = one element of the trajectory): read lane-per-row they would put every lane in its own
                // cache line (72-byte rows), and with four trajectories per CU the address unit, not the memory, sets the pace (measured: 29 us of a 255 us
                // solve).  So blocks of 64 rows are fetched with fully coalesced loads (lane + 64 k), parked in the pool - free until the first command - and each
                // lane then reads ITS row from LDS.  Same sums in the same order as oc_predict_kernel: the two starts are bit-identical.
                constexpr int SB = 64 / NX;                          // stages of the Riccati record per block (lane = (stage, row))
                constexpr int STG = L::PRED_STG;
                static_assert(STG >= 64 * NP && STG >= SB * RSZ, "staging block");
                double* stage = pool;
                double* dxs = pool + STG;                            // dx [t][i], T + 1 nodes
                double dth[NP > 0 ? NP : 1];
#pragma unroll
                for (int j = 0; j < NP; ++j) dth[j] = op.dtheta[(int64_t)b * op.dtheta_bstride + j];
                auto rows_dot = [&](const double* __restrict__ src, int nrows, auto sink) {
                    for (int r0 = 0; r0 < nrows; r0 += 64) {
                        const int nr = min(64, nrows - r0), nd = nr * NP;
                        const double* s_ = src + (int64_t)r0 * NP;
                        double v[NP > 0 ? NP : 1];
#pragma unroll
                        for (int k = 0; k < NP; ++k) { const int idx = lane + 64 * k; v[k] = s_[idx < nd ? idx : 0]; }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < NP; ++k) stage[lane + 64 * k] = v[k];
                        wave_lds_sync();
                        if (lane < nr) {
                            double d = 0.0;
#pragma unroll
                            for (int j = 0; j < NP; ++j) d = fma(stage[lane * NP + j], dth[j], d);
                            sink(r0 + lane, d);
                        }
                        wave_lds_sync();
                    }
                };
                rows_dot(op.dxdp + (int64_t)b * (T + 1) * NX * NP, (T + 1) * NX, [&](int q, double d) {
                    const int t = q / NX, i = q - t * NX;
                    double v = xb[q];
                    if (t > 0) corr_upd(d, v);
                    v += d;
                    s0[i * TS + t] = t == 0 ? x0[(int64_t)b * NX + i] : v;
                    dxs[q] = d;
                });
                rows_dot(op.dudp + (int64_t)b * T * NU * NP, T * NU, [&](int q, double d) {
                    const int t = q / NU, i = q - t * NU;
                    double v = ub[q];
                    corr_upd(d, v);
                    v += d;
                    s0[OU + i * TS + t] = v;
                });
                if (!predl) {
                    for (int q = lane; q < T * NX; q += 64) { const int t = q / NX, i = q - t * NX; s0[OL + i * TS + t] = lb[q]; }
                } else {
                    constexpr int NQR = (SB * RSZ + 63) / 64;
                    wave_lds_sync();
                    for (int t0 = 0; t0 < T; t0 += SB) {
                        const int nst = min(SB, T - t0), nd = nst * RSZ;
                        const double* s_ = op.riccati + ((int64_t)b * T + t0) * RSZ;
                        double v[NQR];
#pragma unroll
                        for (int k = 0; k < NQR; ++k) { const int idx = lane + 64 * k; v[k] = s_[idx < nd ? idx : 0]; }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < NQR; ++k) { const int idx = lane + 64 * k; if (idx < SB * RSZ) stage[idx] = v[k]; }
                        wave_lds_sync();
                        const int sg = lane / NX, i = lane - sg * NX;
                        if (lane < nst * NX) {
                            const int t = t0 + sg;
                            const double* R = stage + sg * RSZ;
                            double d = 0.0;
#pragma unroll
                            for (int j = 0; j < NP; ++j) d = fma(R[NX * NX + i * NP + j], dth[j], d);
#pragma unroll
                            for (int k = 0; k < NX; ++k) d = fma(R[i * NX + k], dxs[(t + 1) * NX + k], d);
                            double v_ = lb[t * NX + i];
                            v_ += d;
                            s0[OL + i * TS + t] = v_;
                        }
                        wave_lds_sync();
                    }
                }
            }
            for (int i = lane; i < NX; i += 64) stp[i * TS] = 0.0;       // dx_0 = 0: x_0 is fixed
        }
        bool dead = false;
        int seq = 0;
#ifdef PDP_MS_TIMING      // timing builds (probes/ms_phase_timing.py): cycles per phase and iteration in the iteration log instead of IPOPT's columns
        long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmi = __builtin_readcyclecounter(), tm0 = tmi;
#define MS2_T0() tm0 = __builtin_readcyclecounter()
#define MS2_T1(k) do { const long long now_ = __builtin_readcyclecounter(); tm[k] += now_ - tm0; tm0 = now_; } while (0)
#else
#define MS2_T0()
#define MS2_T1(k)
#endif
        // commands: parameters first, then the sequence number with release semantics (LDS writes and global stores above are visible to the evaluator)
        int issue_csrc = 0;                      // (1 around the SWEEP of a second-order correction)
        auto issue = [&](int type, double alpha, int cur, int dst) {
            if (lane == 0) { ctl[MS2_TYPE] = type; ctl[MS2_CUR] = cur; ctl[MS2_DST] = dst; ctl[MS2_CSRC] = issue_csrc; res[MS2_ALPHA] = alpha; ctl[MS2_PROD] = 0; ctl[MS2_CONS] = 0; }
            ++seq;
            f3_signal(ctl + MS2_SEQ, seq);
        };
        auto wait_slot = [&](int slot_) {        // (the evaluator has the SIMD's issue slots while the runner has nothing to do)
            __builtin_amdgcn_s_setprio(0);
            if (!dead && !ms2_wait_ge(ctl + slot_, seq, ctl)) dead = true;
            __builtin_amdgcn_s_setprio(3);
        };
        auto wait_done = [&]() { wait_slot(MS2_DONE); };
        auto abort_sweep = [&]() { f3_signal(ctl + MS2_ABORT, seq); wait_done(); };

        const int col = tile_col(lane);
        const auto rsG = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, (int)((int64_t)T * GSZ * 8), 0x00020000);
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)pw, 0, (int)((int64_t)T * PWSZ * 8), 0x00020000);
        // residuals of the current iterate, as the evaluator's last accepted pass left them
        double f_cur = 0.0, th_cur = 0.0, inf_pr = 0.0, inf_du = 0.0, zmax = 0.0, lmax = 0.0, lamc = 0.0;
        bool finite = true;
        auto read_res = [&]() {
            wave_lds_sync();
            f_cur = uni(res[MS2_F]); th_cur = uni(res[MS2_TH]); inf_pr = uni(res[MS2_PR]); inf_du = uni(res[MS2_DU]); zmax = uni(res[MS2_Z]); lmax = uni(res[MS2_L]);
            lamc = uni(res[MS2_LC]);
            finite = uni(res[MS2_FIN]) != 0.0;
        };
        bool PWfinite = true;

        // Backward sweep over the chunks of the current SWEEP command with Hessian scale hs (1; 0 = least-squares multiplier estimate: W = I,
        // no defects) and shift dw.  Returns true when every Quu was positive definite; stops at the first one that is not.
        auto backward = [&](double hs, double dw) -> bool {
            // gather / store maps of this sweep (from an opaque lane id: see opaque())
            const int ln = opaque(lane);
            auto codeS = [](int mat, int i) { return Mdl::sol_code(mat, i); };       // 0 F, 1 G, 2 Hxx, 3 Hxu, 4 Huu
            Gather3 gF, gY, gHxx, gHX, gHU, gGr, gHux;
            if constexpr (AUG) {
                // homogeneous form: F~ = [F c; 0 1], G~ = [G; 0], Hxx~ = [Hxx rx; rx' 0], Hux~ = [Hxu' | ru], Huu
                make_gather3(gF, ln, L::CB0, [&](int r, int c) { return (r < NX && c < NX) ? codeS(0, r * NX + c) : ((r < NX && c == NX) ? L::C0 + r : ((r == NX && c == NX) ? L::ONEB : -1)); });
                // (G~ "by row blocks", one register: lane 16 k + 4 b + i <- G[4 b + k][i]; riccati_backward_aug's second operand form of G)
                make_gather3(gY, ln, L::CB0, [&](int r, int c) { return (r < 4 && (c & 3) < M && 4 * (c >> 2) + r < NX) ? codeS(1, (4 * (c >> 2) + r) * NU + (c & 3)) : -1; });
                make_gather3(gHxx, ln, L::CB0, [&](int r, int c) { return (r < NX && c < NX) ? codeS(2, r * NX + c) : ((r < NX && c == NX) ? L::RX + r : ((r == NX && c < NX) ? L::RX + c : -1)); });
                make_gather3(gHux, ln, L::CB0, [&](int r, int c) { return (r < M && c < NX) ? codeS(3, c * NU + r) : ((r < M && c == NX) ? L::RU + r : -1); });
                make_gather3(gHU, ln, L::CB0, [&](int r, int c) { return (r < M && c < M) ? codeS(4, r * NU + c) : -1; });
                make_gather3(gHX, ln, L::CB0, [&](int r, int c) { return -1; });
            } else {
            make_gather3(gF, ln, L::CB0, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(0, r * NX + (c & 3)) : -1)
                                                                            : ((r < NX && c < NX) ? codeS(0, r * NX + c) : -1); });
            make_gather3(gY, ln, L::CB0, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(1, r * NU + c) : (c == M ? L::C0 + r : -1)); });
            make_gather3(gHux, ln, L::CB0, [&](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? codeS(3, (c & 3) * NU + r) : -1)
                                                                              : ((r < M && c < NX) ? codeS(3, c * NU + r) : -1); });
            make_gather3(gHxx, ln, L::CB0, [&](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? codeS(2, r * NX + (c & 3)) : -1)
                                                                              : ((r < NX && c < NX) ? codeS(2, r * NX + c) : -1); });
            make_gather3(gHX, ln, L::CB0, [&](int r, int c) { return r >= NX ? -1 : (c < M ? codeS(3, r * NU + c) : (c == M ? L::RX + r : -1)); });
            make_gather3(gHU, ln, L::CB0, [&](int r, int c) { return r >= M ? -1 : (c < M ? codeS(4, r * NU + c) : (c == M ? L::RU + r : -1)); });
            }
            make_gather3(gGr, ln, L::CB0, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeS(1, r * NU + (c & 3)) : -1; });
            BufMap mK, mIK, mP, mW;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile_row(ln, r), colb = tile_col(ln);
                mK.voff[r] = (row < NU && colb < NA) ? 8u * (unsigned)L::gK(row, colb) : MS2_OOB;                           // K [NU x NX] (rep form: first column block) / K~ [NU x NA]
                mIK.voff[r] = (row < NU && colb == M) ? 8u * (unsigned)(NX * NU + row) : MS2_OOB;                          // k behind K (not in the homogeneous form)
                mP.voff[r] = (row < NA && colb < NA && (SMALL || row <= colb)) ? 8u * (unsigned)L::pk(row, colb) : MS2_OOB;   // P_{t+1}: full / upper triangle (of P~)
                mW.voff[r] = (row < NX && colb == M) ? 8u * (unsigned)(PSZ + row) : MS2_OOB;
            }
            // per-lane tile masks: diagonal of the n x n / m x m blocks
            d4 dgN, dgM0 = z;
            {
                const int colb = tile_col(ln);
#pragma unroll
                for (int r = 0; r < 4; ++r) dgN[r] = SMALL ? ((r == 0 && (ln >> 4) == (colb & 3) && (colb & 3) < NX) ? 1.0 : 0.0) : ((tile_row(ln, r) == colb && colb < NX) ? 1.0 : 0.0);
                dgM0[0] = ((ln >> 4) == colb && colb < M) ? 1.0 : 0.0;
            }
            const bool scaled = !(hs == 1.0 && dw == 0.0);
            const double sU = col < M ? hs : 1.0, sC = col == M ? hs : 1.0;      // scale of the Hessian columns / of the defect column
            bool pdall = true, ok = true;
            d4 P = z, W2 = z;
            constexpr int RB = 8 * BS;
            MS2_T1(6);                                   // (timing builds: "sweep prologues" = loop top + the maps above)
            for (int g = 0; g < nchunk && pdall && !dead; ++g) {
                int t0, cnt;
                bchunk(g, t0, cnt);
                const double* pb = pool + (g & 1) * L::BUF;
                MS2_T0();
                if (!ms2_wait_ge(ctl + MS2_PROD, g + 1, ctl)) { dead = true; break; }
                MS2_T1(g == 0 ? 0 : 2);
                if (g == 0) {       // terminal stage (the evaluator filled it before the first chunk): P = hs hxx + dw I, W = h_x(x_T) - lambda_T
                    Gather gP;
                    make_gather(gP, lane, L::NCFIN, 0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::fin_code(0, r * NX + (c & 3)) : -1)
                                                                                       : ((r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1); });
                    P = gather_tile(fin, gP, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = tile_row(lane, r);
                        P[r] = hs * P[r] + dw * dgN[r];
                        if constexpr (AUG) {               // P~_T = [hxx g; g' 0], g = h_x(x_T) - lambda_T
                            if (col == NX && row < NX) P[r] = dlT[row];
                            if (row == NX && col < NX) P[r] = dlT[col];
                        } else if (col == M && row < NX) W2[r] = dlT[row];
                    }
                }
                int tl = cnt - 1;
                const double* r0 = pb + (tl - 1) * BS;      // the runs sit one row BELOW the step's row: the step reads at +RB, its one-step-ahead requests at +0
                Run3 rF = run3_at(gF, r0), rY = run3_at(gY, r0), rHxx = run3_at(gHxx, r0), rHX = run3_at(gHX, r0), rHU = run3_at(gHU, r0),
                     rGr = run3_at(gGr, r0), rHux = run3_at(gHux, r0);
                constexpr int NRY = AUG ? 1 : NRT;              // (homogeneous form: G by row blocks, one register)
                auto move_all = [&](int bytes) {
                    move3<NRT>(rF, bytes); move3<NRY>(rY, bytes); move3<NRT>(rHxx, bytes); move3<NRT>(rHX, bytes); move3<1>(rHU, bytes); move3<NRT>(rGr, bytes); move3<1>(rHux, bytes);
                };
                d4 Fa = read3<NRT>(rF, RB), Ya = read3<NRY>(rY, RB), Fb = z, Yb = z;
                auto bstep = [&](int tl, unsigned imm, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {
                    const int t = t0 + tl;
                    d4 Hxx = read3<NRT>(rHxx, imm), HX2 = z, HU2 = read3<1>(rHU, imm), Grep = read3<NRT>(rGr, imm), Hux = read3<1>(rHux, imm);
                    if constexpr (!AUG) HX2 = read3<NRT>(rHX, imm);
                    if (tl > 0) { Fn = read3<NRT>(rF, imm - RB); Yn = read3<NRY>(rY, imm - RB); }
                    d4 Ys = Yc, Fs = Fc;
                    double Hux0 = Hux[0];
                    if (scaled) {
                        if constexpr (AUG) {               // Hessian blocks x hs (+ dw I), the gradient row / column and the 1 of F~ as they are, the defect column x hs
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool blk_ = tile_row(lane, r) < NX && col < NX;
                                Hxx[r] = blk_ ? hs * Hxx[r] + dw * dgN[r] : Hxx[r];
                                Fs[r] = (col == NX && tile_row(lane, r) < NX) ? hs * Fc[r] : Fc[r];
                            }
                            HU2[0] = hs * HU2[0] + dw * dgM0[0];
                            Hux0 = col < NX ? hs * Hux0 : Hux0;
                        } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { Hxx[r] = hs * Hxx[r] + dw * dgN[r]; HX2[r] *= sU; }
                        HU2[0] = sU * HU2[0] + dw * dgM0[0];
                        Ys = Yc * sC;
                        Hux0 = hs * Hux0;
                        }
                    }
                    // P_{t+1}, W_{t+1} (homogeneous form: P~_{t+1}): the multiplier step dlam_t = P_{t+1} dx_{t+1} + W_{t+1} of the forward sweep needs them
                    const unsigned soP = (unsigned)(t * PWSZ) * 8u, soG = (unsigned)(t * GSZ) * 8u;
                    buf_store<NRT>(rsP, soP, mP, P);
                    if constexpr (!AUG) buf_store<NRT>(rsP, soP, mW, W2);
                    if constexpr (AUG) {
                        RiccatiGains gn;
                        ok = riccati_backward_aug<M, true>(P, Fs, Ys[0], Grep, Hxx, HU2[0], Hux0, scratch, lane, gn) && ok;
                        pdall = pdall && gn.pd;
                        buf_store<1>(rsG, soG, mK, gn.K);
                    } else if constexpr (SMALL) {
                        SmallGains gs;
                        double Pr = P[0], Wr = W2[0];
                        ok = riccati_small_backward<M, true>(Pr, Wr, Fc[0], Ys[0], Grep[0], Hxx[0], HX2[0], HU2[0], Hux0, lane, tlane, 1, gs) && ok;
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
                MS2_T1(1);
            }
            pdall = pdall && ok && !dead;
            PWfinite = pdall ? __all(tile_finite(P) && tile_finite(W2)) : true;      // (an aborted sweep leaves P, W undefined)
            return pdall;
        };

        // Forward pass of the LQ problem over the forward chunks: dx, du (homogeneous form: and dlam) into the workspace; in the other forms the evaluator follows with dlam;
        // returns grad(phi)' d = grad(L)' d + lambda' c  (A d = -c)
        auto forward = [&](double hs) -> double {
            const int ln = opaque(lane);
            BufMap mKT, mPld, mIK;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile_row(ln, r), colf = tile_col(ln);
                mIK.voff[r] = (row < NU && colf == M) ? 8u * (unsigned)(NX * NU + row) : MS2_OOB;                           // k behind K (not in the homogeneous form)
                mKT.voff[r] = (row < NA && (colf & 3) < NU) ? 8u * (unsigned)L::gK(colf & 3, row) : MS2_OOB;                // K (K~) read back transposed, replicated in the column blocks
                mPld.voff[r] = (AUG && row < NA && colf < NA) ? 8u * (unsigned)L::pk(row, colf) : MS2_OOB;                   // P~ read back as a full tile from its upper triangle
            }
            Gather3 gFT, gGT, gE, gRX, gRU;
            make_gather3(gFT, ln, L::CF0, [](int r, int c) { return SMALL ? ((r < NX && (c & 3) < NX) ? Mdl::solf_code(0, (c & 3) * NX + r) : -1)
                                                                          : ((r < NX && c < NX) ? Mdl::solf_code(0, c * NX + r)
                                                                             : (AUG && r == NX && c < NX ? L::FC0 + c : (AUG && r == NX && c == NX ? L::ONEF : -1))); });      // F~' = [F' 0; c' 1]
            make_gather3(gGT, ln, L::CF0, [](int r, int c) { return SMALL ? ((r < M && (c & 3) < NX) ? Mdl::solf_code(1, (c & 3) * NU + r) : -1)
                                                                          : ((r < M && c < NX) ? Mdl::solf_code(1, c * NU + r) : -1); });
            make_gather3(gE, ln, L::CF0, [](int r, int c) { return (r < NX && c == M) ? L::FC0 + r : -1; });
            make_gather3(gRX, ln, L::CF0, [](int r, int c) { return (r < NX && c == M) ? L::FRX + r : -1; });
            make_gather3(gRU, ln, L::CF0, [](int r, int c) { return (r < M && c == M) ? L::FRU + r : -1; });
            BufMap mDX, mDU;                                 // dx_{t+1} / du_t: column M of the tile, row i to [i TS + stage]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile_row(lane, r);
                mDX.voff[r] = (col == M && row < NX) ? 8u * (unsigned)(row * TS) : MS2_OOB;
                mDU.voff[r] = (col == M && row < NU) ? 8u * (unsigned)(OU + row * TS) : MS2_OOB;
            }
            const auto rsD = __builtin_amdgcn_make_buffer_rsrc((void*)stp, 0, (int)(GRP * 8), 0x00020000);
            // feedback gains of stage t are requested two steps ahead (three register sets in rotation)
            struct Gn { d4 KT, k, Pt; };                     // Pt (homogeneous form): P~_{t+1} as a full tile, for the multiplier step below
            // (loaded as stored, +K and +k: a negation right behind the load would make the step wait for the loads it has just issued; the sign is
            // absorbed in the products below: V = K x + k = -du, dx+ = F dx + c - G V)
            auto ldg = [&](int t) {
                const int tt = t < T ? t : T - 1;
                Gn s;
                const unsigned so = (unsigned)(tt * GSZ) * 8u;
                s.KT = buf_load<NRT>(rsG, so, mKT);
                s.k = z;
                s.Pt = z;
                if constexpr (!AUG) s.k = buf_load<1>(rsG, so, mIK);
                if constexpr (AUG) s.Pt = buf_load<4>(rsP, (unsigned)(tt * PWSZ) * 8u, mPld);
                return s;
            };
            // gains of the next stages: requested two stages ahead into the set that is free, or - small systems, whose step is short against the latency of
            // the workspace - three stages ahead, a step refilling the set it has just used (the tile forms measured slower that way:
            // profiles/r03_ms2_variants.txt)
            constexpr bool AHEAD3 = SMALL;
            Gn A = ldg(0), Bn = ldg(1), Cn;
            if constexpr (AHEAD3) Cn = ldg(2);
            else { Cn.KT = z; Cn.k = z; Cn.Pt = z; }
            d4 X2 = z, Xb = z;
            if constexpr (AUG) {
                // x~_0 = [dx_0 = 0; 1], carried in EVERY column of the tile (the products below are column-wise, so every column stays x~_t): as the left
                // operand of the four-row product the tile then is x~ "replicated in the column blocks", and the multiplier step
                //     dlam_t' = x~_{t+1}' P~_{t+1}      (first NX entries; PDP.py:604)
                // costs 4 small MFMAs on the runner, off its dependency chain - against a 16x16x16 product in round 2, or a lane-per-stage pass on the
                // evaluator whose (P, W) reads arrived as one burst at the end of the sweep (profiles/r03_ms2_variants.txt)
#pragma unroll
                for (int r = 0; r < 4; ++r) X2[r] = tile_row(lane, r) == NX ? 1.0 : 0.0;
            }
            BufMap mDL;                                      // dlam_t: row 0 of the four-row product, entry c to [OL + c TS + t]
#pragma unroll
            for (int r = 0; r < 4; ++r) mDL.voff[r] = (r == 0 && lane < NX) ? 8u * (unsigned)(OL + lane * TS) : MS2_OOB;
            double acc = 0.0;
            const bool scaledE = hs != 1.0;
            constexpr int RF = 8 * FS;
            MS2_T1(6);
            for (int c = 0; c < nchunkF && !dead; ++c) {
                const int g = nchunk + c;
                int t0, cnt;
                fchunk(c, t0, cnt);
                const double* pb = pool + (g & 1) * L::BUF;
                MS2_T0();
                if (!ms2_wait_ge(ctl + MS2_PROD, g + 1, ctl)) { dead = true; break; }
                MS2_T1(2);
                Run3 rFT = run3_at(gFT, pb), rGT = run3_at(gGT, pb), rE = run3_at(gE, pb), rRX = run3_at(gRX, pb), rRU = run3_at(gRU, pb);
                auto move_all = [&](int bytes) { move3<NRT>(rFT, bytes); move3<1>(rGT, bytes); move3<NRT>(rE, bytes); move3<NRT>(rRX, bytes); move3<1>(rRU, bytes); };
                auto fstep = [&](int tl, unsigned imm, const d4 Xc, d4& Xn, Gn& cur, Gn& fill) {
                    const int t = t0 + tl;
                    if constexpr (!AHEAD3) fill = ldg(t + 2);
                    d4 FT = read3<NRT>(rFT, imm);
                    d4 GT = read3<1>(rGT, imm);
                    d4 E2 = z;
                    if constexpr (!AUG) E2 = read3<NRT>(rE, imm);
                    d4 RXn = read3<NRT>(rRX, imm);
                    d4 RUc = read3<1>(rRU, imm);
                    if (scaledE) {
                        if constexpr (AUG) {                // the defect sits in row NX of F~'
#pragma unroll
                            for (int r = 0; r < 4; ++r) FT[r] = (tile_row(lane, r) == NX && col < NX) ? hs * FT[r] : FT[r];
                        } else E2 = E2 * hs;
                    }
                    d4 U2 = z;
                    if constexpr (SMALL) {
                        Xn = z;
                        const double v0 = mma4_blk(cur.KT[0], Xc[0], cur.k[0]);         // K dx + k = -du
                        double x1 = mma4_blk(FT[0], Xc[0], E2[0]);                      // F dx + c
                        x1 = mma4_blk(-GT[0], v0, x1);                                  // - G (K dx + k)
                        U2[0] = -v0; Xn[0] = x1;
                        acc += RXn[0] * x1 + RUc[0] * U2[0];
                    } else {
                        d4 V = z;
                        V[0] = mma4_tn(cur.KT, Xc, AUG ? 0.0 : cur.k[0]);      // K~ x~ = K dx + k
                        Xn = mma_tn(FT, Xc, AUG ? z : E2);                     // F~ x~ = [F dx + c; 1]
                        Xn = mms_tn_r0(GT, V, Xn);
                        U2[0] = -V[0];
                        acc += RXn[0] * Xn[0] + RXn[1] * Xn[1] + RXn[2] * Xn[2] + RXn[3] * Xn[3] + RUc[0] * U2[0];
                        if constexpr (AUG) {
                            d4 Lm = z;
                            Lm[0] = mma4_tn(Xn, cur.Pt, 0.0);          // x~_{t+1}' P~_{t+1}: rows 0..3 identical, row 0 = [dlam_t' | .]
                            buf_store<1>(rsD, (unsigned)t * 8u, mDL, Lm);
                        }
                    }
                    buf_store<1>(rsD, (unsigned)t * 8u, mDU, U2);
                    buf_store<NRT>(rsD, (unsigned)(t + 1) * 8u, mDX, Xn);
                    if constexpr (AHEAD3) cur = ldg(t + 3);
                };
                int tl = 0;
                for (; tl + 3 <= cnt; tl += 3) {
                    fstep(tl, 0u, X2, Xb, A, Cn);
                    fstep(tl + 1, (unsigned)RF, Xb, X2, Bn, A);
                    fstep(tl + 2, (unsigned)(2 * RF), X2, Xb, Cn, Bn);
                    X2 = Xb;
                    move_all(3 * RF);
                }
                for (; tl < cnt; ++tl) { fstep(tl, 0u, X2, Xb, A, Cn); X2 = Xb; const Gn nx = A; A = Bn; Bn = Cn; Cn = nx; move_all(RF); }
                f3_signal(ctl + MS2_CONS, g + 1);       // release: dx, du of the chunk are in memory
                MS2_T1(3);
            }
            return ms2_sum(acc) + lamc;
        };

        // ---- main loop (IPOPT's order: convergence test, sweep with inertia correction, line search).
        // phase 0 (cold start only): the least-squares multiplier estimate (constr_mult_init_max = 1000),
        //     [I A'; A 0] [w; lambda] = -[grad f; 0]  - the same sweep with W = I and no defects; phase 1: the iteration.
        // PDP_MS_FROM_CONTROLS (with PDP_MS_WARM): only the caller's controls count - the states become their rollout, the multipliers the least-squares
        // estimate (the RESTORE command, then phase 0 like a cold start)
        const bool from_u = warm && (op.flags & PDP_MS_FROM_CONTROLS) != 0;
        const bool ph1 = warm && !from_u;
        int st = 0, it = 0, nfilt = 0, conv = 0, phase = ph1 ? 1 : 0, cur = 0;
        double hs = ph1 ? 1.0 : 0.0, dw = ph1 ? 0.0 : 1.0, dw_last = 0.0, theta_max = 0.0, theta_min = 0.0;
        bool gains_ok = false, pending = false;         // pending: the evaluator is already on the sweep of the current iterate (TRIAL_SWEEP)
        // PDP_MS_PREDICT_GUARD (round 5): a first-order prediction is only as good as the linearisation it comes from.  On the reference's own rocket IRL run (stored trace,
        // row 1: a parameter step of 1 % where the sensitivities are of order 1e2) the predicted point has FIFTY times the KKT error of the previous solution it was meant
        // to improve, and Newton's method started there ends in another stationary point (loss 10289.86 where IPOPT stored 1301.24; oracle/ipopt_ms.py reproduces both).
        // So the previous solution itself - the plain warm start - is loaded into point set 1 and evaluated first (one residual pass, ~5 % of a two-iteration solve);
        // the predicted point in set 0 is evaluated next, with the first sweep speculated behind it as always; the prediction is kept iff its scaled KKT error
        //     max(inf_pr / (1 + max|x|,|u|), inf_du / (1 + max|lam|))        (the quantities of the convergence test)
        // is finite and not larger than the plain point's.  Otherwise the speculative sweep is dropped and the iteration starts from set 1 (PDP_MS_PREDICT_REJECTED).
        // A prediction of states and controls only (PDP_MS_PREDICT_PRIMAL, or no Riccati record) is judged by the primal part alone: both candidates carry the SAME
        // multipliers, so the dual residual says nothing about the quality of the prediction (at a 2 % step it is a coin flip that would throw away predictions which
        // save an iteration - oracle: cart-pole and quadrotor demos, primal infeasibility 1e-4 against 3e-2, dual 5.3 against 5.1).
        // A correction that moves no state or control by more than PDP_MS_GUARD_TRUST (2 %) of max(1, |its value|) is trusted without the extra pass: the error of a
        // first-order prediction is of the order of the square of that.  (The steps of a running gradient-descent loop are ~1e-4: the guard then costs nothing; the
        // rocket's rejected predictions move the trajectory by 150 %.)
        const double corr = ms2_max(corr_l);
        const bool guard = ph1 && (rec || pred) && (op.flags & PDP_MS_PREDICT_GUARD) != 0 && !(corr <= PDP_MS_GUARD_TRUST);
        const bool g_primal = rec ? recp : !predl;
        double g_f = 0.0, g_th = 0.0, g_pr = 0.0, g_du = 0.0, g_z = 0.0, g_l = 0.0, g_lc = 0.0, g_err = 0.0;
        bool g_fin = false;
        // residuals of the starting point (both phases: the phase-0 sweep takes its right-hand sides from the same arrays).  Unless the point has to be built first
        // (RESTORE), the evaluator goes straight on with the first sweep at it (TRIAL_SWEEP with alpha = 0, source = destination): the runner reads the residuals when
        // the trial half is done and finds chunk 0 of the sweep already under way instead of asking for it then
        if (from_u) issue(MS2_CMD_RESTORE, 0.0, cur, cur);
        else if constexpr (SPLIT0) issue(MS2_CMD_TRIAL, 0.0, cur, cur);      // (small systems: a TRIAL_SWEEP's pass is shared between the two waves - the line search only)
        else issue(MS2_CMD_TRIAL_SWEEP, 0.0, cur, cur);
        if (guard) {
            // The previous solution has been evaluated by the EVALUATOR while this wave was forming the prediction (see its prologue: the pass starts at kernel entry, on
            // the strength of the flags alone - whether the correction is small enough to be trusted unseen is not known before the prediction exists; a pass nobody asks
            // for costs an idle wave nothing).  Rounds 5 / 6 ran it behind, then beside the pass over the predicted point: +13 / +10 us per solve at C3.
            __builtin_amdgcn_s_setprio(0);
            if (!ms2_wait_ge(ctl + MS2_GDONE, 1, ctl)) dead = true;
            __builtin_amdgcn_s_setprio(3);
            wave_lds_sync();
            g_f = uni(res[MS2_G0 + 0]); g_th = uni(res[MS2_G0 + 1]); g_pr = uni(res[MS2_G0 + 2]); g_du = uni(res[MS2_G0 + 3]); g_z = uni(res[MS2_G0 + 4]); g_l = uni(res[MS2_G0 + 5]);
            g_lc = uni(res[MS2_G0 + 6]); g_fin = uni(res[MS2_G0 + 7]) != 0.0;
            g_err = g_primal ? g_pr / (1.0 + g_z) : fmax(g_pr / (1.0 + g_z), g_du / (1.0 + g_l));
        }
        if (from_u || SPLIT0) wait_done();
        else { wait_slot(MS2_TDONE); pending = true; }
        read_res();
        if (guard && !dead) {
            const double p_err = g_primal ? inf_pr / (1.0 + zmax) : fmax(inf_pr / (1.0 + zmax), inf_du / (1.0 + lmax));
            if (g_fin && !(finite && p_err <= g_err)) {            // the prediction is no improvement (or not finite): start from the previous solution
                if (pending) { abort_sweep(); pending = false; }
                cur = 1;
                f_cur = g_f; th_cur = g_th; inf_pr = g_pr; inf_du = g_du; zmax = g_z; lmax = g_l; lamc = g_lc; finite = true;
                st |= PDP_MS_PREDICT_REJECTED;
            }
        }
        // Second-order correction (Waechter & Biegler 2006, section 2.4; IPOPT's defaults max_soc = 4, kappa_soc = 0.99): when the FIRST trial point of an iteration
        // (alpha = 1: there are no bounds) is rejected and its constraint violation is not below the iterate's, the step is corrected by solving the SAME KKT matrix
        // with the constraint block c_soc = alpha c(x_k) + c(x_k + alpha d), accumulated over up to four attempts while each reduces theta to 99 % of the attempt
        // before; a corrected point is tested with the ORIGINAL alpha and directional derivative.  IPOPT re-uses its factorisation for that; here the matrices are
        // not kept (the sweep stores gains and P only), so a correction is one more SWEEP of the iterate whose defect column the evaluator takes from `csoc`
        // (MS2_CSRC) - a full Newton sweep for what is a back-substitution in IPOPT, on a path that only cold solves far from their optimum take.  The plain
        // step waits in `stpb` and comes back when the corrections fail.  oracle/ipopt_ms.py: solve(soc=True) is the same algorithm.  Off unless PDP_MS_WITH_SOC
        // (include/pdp_hip.h says why).
        // What the line search must remember across the sweep of a correction (directional derivative, alpha_min, theta of the attempt before, the two counters) waits
        // in the mailbox, not in registers: held live across the sweeps, those few values were exactly what the 256-register instantiation then spilled.
        const bool soc_on = (op.flags & PDP_MS_WITH_SOC) != 0;
        // Watchdog (IPOPT's BacktrackingLineSearch: watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3; PDP_MS_WITH_WATCHDOG, instantiations WD only).  After ten
        // consecutive iterations whose accepted step was shortened (alpha < 1) the iterate and its direction are stored; for up to three iterations the FULL step is then
        // taken whether acceptable or not, each trial point tested against the STORED point's (theta, phi, grad(phi)'d) and the filter; the first acceptable one ends the
        // procedure (the filter is augmented from the stored point's values), otherwise the stored iterate and direction come back and the regular search continues from
        // alpha = 1/2.  oracle/ipopt_ms.py: solve(watchdog=True) is the same, restated from memory of IPOPT's structure - no IPOPT here to pin it on, hence opt-in; why it
        // exists at all: profiles/r06_solver_iterlog_stats.txt (cold rocket solves at T = 100 crawl with steps of 1e-3 for hundreds of iterations; the trigger is met in 199
        // of 256) and profiles/r06_watchdog_experiment.txt (the restatement with it: 15 of 16 within 300 iterations instead of 12, the same optima).
        const bool wd_on = WD && (op.flags & PDP_MS_WITH_WATCHDOG) != 0;
        bool in_wd = false;
        int wd_short = 0, wd_trial = 0;
        double w_f = 0.0, w_th = 0.0, w_gd = 0.0, w_dw = 0.0, w_amin = 0.0, w_pr = 0.0, w_du = 0.0;
        // A free watchdog trial may land where the iteration cannot go on (not finite, or no inertia correction below 1e20 succeeds: 33 of 512 cold rocket solves at
        // T = 100 ended that way before this existed): the procedure ends, the stored iterate comes back, its residuals are evaluated again, and the iteration goes on
        // from there as a regular one (the direction is computed anew: the regularisation history has moved on).  One log row (alpha = 0) and one iteration are spent.
        auto wd_fallback = [&](double& dw_, int cur_) {
            __threadfence_block();
            double* __restrict__ sp = Pt(cur_);
            for (int q = lane; q < (int)GRP; q += 64) sp[q] = wdp[q];
            __threadfence_block();
            in_wd = false; wd_short = 0;
            issue(MS2_CMD_TRIAL, 0.0, cur_, cur_);
            wait_done();
            read_res();
#ifndef PDP_MS_TIMING
            if (iter_log && it < op.log_rows && lane == 0) {
                double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
                row[0] = it; row[1] = w_f; row[2] = w_pr; row[3] = w_du; row[4] = dw_; row[5] = 0.0; row[6] = w_gd; row[7] = w_th;
            }
#endif
            dw_ = 0.0;
            gains_ok = false;
            ++it;
        };
#ifdef PDP_MS_TIMING
        const long long tk1 = tmi, tk2 = __builtin_readcyclecounter();
#endif
        for (;;) {
            if (dead) break;
            if constexpr (WD) {
                if (in_wd && phase == 1 && dw == 0.0 && !finite) {      // a free watchdog trial that is not finite on the dual side
                    if (pending) { abort_sweep(); pending = false; }
                    if (dead) break;
                    wd_fallback(dw, cur);
                    continue;
                }
            }
            const int soc_sweep = uni(ctl[MS2_SOCM]);   // this pass of the loop computes a correction of the step, not a step
            bool soc_fail = false;
            if (phase == 1 && dw == 0.0 && !soc_sweep) { // a new iterate: converged?  (a sweep follows only if not - or once more for the gains output)
                bool stop_ = false;
                if (!finite) { st |= PDP_STATUS_NONFINITE; stop_ = true; }
                else {
                    if (it == 0) { theta_max = 1e4 * fmax(1.0, th_cur); theta_min = 1e-4 * fmax(1.0, th_cur); }
                    if (inf_pr <= op.tol * (1.0 + zmax) && inf_du <= op.tol * (1.0 + lmax)) { conv = 1; if (!gains_out) stop_ = true; }
                    else if (it >= op.max_iter) { st |= PDP_MS_MAXITER; stop_ = true; }
                }
                if (stop_) { if (pending) abort_sweep(); break; }
            }
            if (!pending) { issue_csrc = soc_sweep ? 1 : 0; issue(MS2_CMD_SWEEP, 0.0, cur, cur); issue_csrc = 0; }
            pending = false;
            const bool pd = backward(hs, dw);
            if (dead) break;
            if (conv || !pd || (phase == 1 && !PWfinite)) {         // the sweep ends here: tell the evaluator to drop the remaining chunks
                abort_sweep();
                if (dead) break;
            }
            gains_ok = pd && PWfinite;
            if (conv) break;                            // (the sweep at the solution left the LQR gains in the workspace)
            if (phase == 0) {
                if (!(pd && PWfinite)) {
                    if (pd) { abort_sweep(); if (dead) break; }
                    phase = 1; hs = 1.0; dw = 0.0; continue;
                }
            } else if (soc_sweep) {
                soc_fail = !(pd && PWfinite);           // (the matrix of a step that has just been accepted by the inertia test: cannot happen short of a non-finite c_soc)
            } else {
                if (!PWfinite) {
                    if constexpr (WD) { if (in_wd) { wd_fallback(dw, cur); continue; } }
                    st |= PDP_STATUS_NONFINITE; break;
                }
                if (!pd) {                              // Algorithm IC (defaults: 1e-4 first, x100 / x8 up, /3 down, 1e20 max)
                    if (dw == 0.0) dw = dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last * (1.0 / 3.0));
                    else dw *= dw_last == 0.0 ? 100.0 : 8.0;
                    if (dw > 1e20) {
                        if constexpr (WD) { if (in_wd) { wd_fallback(dw, cur); continue; } }
                        st |= PDP_MS_INERTIA; break;
                    }
                    continue;
                }
                if (dw > 0.0) dw_last = dw;
            }
            double gd = 0.0;
            if (!soc_fail) {
                gd = forward(hs);
                MS2_T0();
                wait_done();                            // the evaluator is through with the sweep (forms other than the homogeneous one: it has finished dlam)
                MS2_T1(4);
                if (dead) break;
            }
            if (phase == 0) {
                double lm = 0.0;
                bool fin = true;
                const double* dl_ = stp + OL;
                for (int q = lane; q < NX * TS; q += 64) if (q % TS < T) { const double v = dl_[q]; lm = fmax(lm, fabs(v)); fin = fin && fabs(v) <= 1.7e308; }
                lm = ms2_max(lm);
                if (__all(fin) && lm <= 1000.0) {
                    double* lc_ = Pt(cur) + OL;
                    for (int q = lane; q < NX * TS; q += 64) if (q % TS < T) lc_[q] = dl_[q];
                    issue(MS2_CMD_TRIAL, 0.0, cur, cur);        // the residuals change with the multipliers
                    wait_done();
                    read_res();
                }
                phase = 1; hs = 1.0; dw = 0.0;
                continue;
            }
            double f = f_cur, theta = th_cur;
            // backtracking filter line search (Algorithm A): alpha_min below which IPOPT would enter the restoration phase
            int soc_mode = soc_sweep, soc_n = 0;
            double alpha = 1.0, amin = 1e-5, soc_th_old = 0.0;
            auto step_back = [&]() {                    // the corrections failed: the plain step again, halved
                __threadfence_block();
                for (int q = lane; q < (int)GRP; q += 64) stp[q] = stpb[q];
                __threadfence_block();
                soc_mode = 0;
                if (lane == 0) ctl[MS2_SOCM] = 0;
                alpha *= 0.5;
            };
            if (!soc_mode) {
                if (gd < 0.0) {
                    amin = fmin(1e-5, 1e-8 * theta / (-gd));
                    if (theta <= theta_min) amin = fmin(amin, pow(theta, 1.1) / pow(-gd, 2.3));
                }
                amin *= 0.05;
            } else {                                    // (a corrected step is judged with the directional derivative of the plain one; alpha = 1: corrections start there)
                wave_lds_sync();
                gd = uni(res[MS2_S_GD]); amin = uni(res[MS2_S_AMIN]); soc_th_old = uni(res[MS2_S_THOLD]); soc_n = uni(ctl[MS2_SOCN]);
                if (soc_fail) step_back();
            }
            double ft = 0.0, tht = 0.0;
            bool accepted = false, ftype = false, fin_p = true, to_soc = false;
            bool wd_free = false;                       // a watchdog trial taken although it is not acceptable
            double rf = f, rth = theta;                 // the point the accepting test referred to (the filter is augmented from it)
            if constexpr (WD) {
                if (wd_on && !in_wd && !soc_mode && wd_short >= 10) {        // start (once per iteration, not in the pass of a correction): the iterate into wdp, its direction into wds
                    __threadfence_block();
                    const double* __restrict__ sp = Pt(cur);
                    for (int q = lane; q < (int)GRP; q += 64) { wdp[q] = sp[q]; wds[q] = stp[q]; }
                    __threadfence_block();
                    in_wd = true; wd_trial = 0;
                    w_f = f; w_th = theta; w_gd = gd; w_dw = dw; w_amin = amin; w_pr = inf_pr; w_du = inf_du;
                    st |= PDP_MS_WATCHDOG;
                }
            }
            while (soc_mode || (alpha >= amin && alpha > 1e-300)) {      // (the second bound only guards against amin = 0)
                const double a_try = soc_mode ? 1.0 : alpha;           // (a corrected step is taken in full: no bounds, no fraction-to-the-boundary rule)
                issue(MS2_CMD_TRIAL_SWEEP, a_try, cur, cur ^ 1);
                if constexpr (SPLIT) {
                    trial_pass(PartPrimal{}, a_try, cur, cur ^ 1); // this wave's half: (theta, phi) of the trial point - all the filter asks for
                    fin_p = fin_all;
                    f3_signal(ctl + MS2_PDONE, seq);               // release: trial (x, u) and defects are in memory
                    ft = a_f; tht = a_th;
                } else {
                    wait_slot(MS2_TDONE);
                    if (dead) break;
                    wave_lds_sync();
                    ft = uni(res[MS2_F]); tht = uni(res[MS2_TH]);
                }
                bool okf = fabs(ft) <= 1.7e308 && fabs(tht) <= 1.7e308 && tht <= theta_max;
                if (okf) {
                    bool dominated = false;
                    for (int e = lane; e < nfilt; e += 64) dominated = dominated || (tht >= fth[e] && ft >= fph[e]);
                    okf = !__any(dominated);
                }
                double rgd = gd;
                rf = f; rth = theta;
                if constexpr (WD) { if (in_wd) { rf = w_f; rth = w_th; rgd = w_gd; } }      // (a watchdog trial is judged from the stored point)
                if (okf) {
                    const bool switching = rgd < 0.0 && alpha * pow(-rgd, 2.3) > pow(rth, 1.1);
                    if (rth <= theta_min && switching) {
                        if (ft <= rf + 1e-8 * alpha * rgd + 10.0 * 2.220446049250313e-16 * fabs(rf)) { accepted = true; ftype = true; }
                    } else if (tht <= (1.0 - 1e-5) * rth || ft <= rf - 1e-8 * rth) accepted = true;
                }
                if constexpr (WD) {
                    if (in_wd) {
                        if (accepted) { in_wd = false; wd_short = 0; }                                  // the procedure succeeded
                        else {
                            ++wd_trial;
                            if (fabs(ft) <= 1.7e308 && fabs(tht) <= 1.7e308 && wd_trial <= 3) { accepted = true; wd_free = true; }      // taken all the same
                            else {                      // failed: back to the stored iterate, the regular search along its direction, the full step known to fail
                                abort_sweep();
                                if (dead) break;
                                __threadfence_block();
                                double* __restrict__ sp = Pt(cur);
                                for (int q = lane; q < (int)GRP; q += 64) { sp[q] = wdp[q]; stp[q] = wds[q]; }
                                __threadfence_block();
                                f = w_f; theta = w_th; gd = w_gd; dw = w_dw; amin = w_amin; inf_pr = w_pr; inf_du = w_du;
                                in_wd = false; wd_short = 0;
                                alpha = 0.5;
                                continue;
                            }
                        }
                    }
                }
                if (accepted) break;
                abort_sweep();                          // (the evaluator went on with the sweep of the rejected point)
                if (dead) break;
                const bool fin_t = fabs(ft) <= 1.7e308 && fabs(tht) <= 1.7e308;
                const double* __restrict__ ctr = Rs(cur ^ 1) + OL;      // defects of the rejected trial point
                if (!soc_mode) {
                    if (soc_on && alpha == 1.0 && fin_t && tht >= theta) {      // A-5.5: first trial point, no progress towards feasibility
                        const double* __restrict__ cc_ = Rs(cur) + OL;
                        __threadfence_block();
                        for (int q = lane; q < (int)GRP; q += 64) stpb[q] = stp[q];
                        for (int q = lane; q < NX * TS; q += 64) if (q % TS < T) csoc[q] = fma(alpha, cc_[q], ctr[q]);
                        __threadfence_block();
                        if (lane == 0) { ctl[MS2_SOCM] = 1; ctl[MS2_SOCN] = 0; res[MS2_S_GD] = gd; res[MS2_S_AMIN] = amin; res[MS2_S_THOLD] = tht; }
                        to_soc = true;
                        break;
                    }
                    alpha *= 0.5;
                } else {
                    ++soc_n;
                    if (soc_n < 4 && fin_t && tht <= 0.99 * soc_th_old) {      // A-5.9, A-5.10: the next correction accumulates on this one
                        __threadfence_block();
                        for (int q = lane; q < NX * TS; q += 64) if (q % TS < T) csoc[q] = fma(1.0, csoc[q], ctr[q]);
                        __threadfence_block();
                        if (lane == 0) { ctl[MS2_SOCN] = soc_n; res[MS2_S_THOLD] = tht; }
                        to_soc = true;
                        break;
                    }
                    step_back();
                }
            }
            if (dead) break;
            if (to_soc) continue;                       // (nothing pending: the sweep of the correction is asked for at the top)
            const bool soc_taken = accepted && soc_mode != 0;
            if (soc_mode && lane == 0) ctl[MS2_SOCM] = 0;
            if (soc_taken) st |= PDP_MS_SOC;
            pending = accepted;
            MS2_T1(5);
#ifndef PDP_MS_TIMING
            if (iter_log && it < op.log_rows && lane == 0) {
                double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
                row[0] = it; row[1] = f; row[2] = inf_pr; row[3] = inf_du; row[4] = dw; row[5] = accepted ? (soc_taken ? -alpha : alpha) : 0.0; row[6] = gd; row[7] = theta;      // (a corrected step: minus its test step length)
            }
#endif
            if (!accepted) {                            // (every rejected trial's sweep has been aborted above)
                // alpha < alpha_min: IPOPT switches to its feasibility restoration phase.  What the filter method needs from that phase is a point with a smaller
                // constraint violation that the filter accepts; multiple shooting offers one directly - keep the controls, replace the states by their rollout
                // from x_0 (theta = 0) - and as in IPOPT the current point joins the filter first and the multipliers are reset to the least-squares estimate
                // afterwards (phase 0 again).  oracle/ipopt_ms.py does the same, and reaches IPOPT's stored optimum this way on the one stored demo that gets
                // here (robot arm demo 3).  Not available: at a feasible point (nothing to restore), with PDP_MS_NO_RESTORATION, or when the rollout overflows.
                if (!(theta > 0.0) || (op.flags & PDP_MS_NO_RESTORATION)) { st |= PDP_MS_RESTORATION; break; }
                if (lane == 0) { fth[nfilt] = (1.0 - 1e-5) * theta; fph[nfilt] = f - 1e-8 * theta; }
                ++nfilt;
                __threadfence_block();
                wd_short = 0;
                issue(MS2_CMD_RESTORE, 0.0, cur, cur);
                wait_done();
                if (dead) break;
                read_res();
                if (!finite) { st |= PDP_MS_RESTORATION; break; }
                st |= PDP_MS_RESTORED;
                phase = 0; hs = 0.0; dw = 1.0; gains_ok = false;
                ++it;
                continue;
            }
            if (!ftype && !wd_free) {                   // (at most one entry per iteration: the workspace holds max_iter + 1; a free watchdog trial adds none)
                if (lane == 0) { fth[nfilt] = (1.0 - 1e-5) * rth; fph[nfilt] = rf - 1e-8 * rth; }
                ++nfilt;
                __threadfence_block();
            }
            if constexpr (WD) { if (!in_wd) wd_short = (alpha == 1.0 || soc_taken) ? 0 : wd_short + 1; }       // consecutive shortened iterations: the watchdog's trigger
            // the accepted trial's residuals are the new iterate's: the primal side from this wave's half of the pass, the dual side from the evaluator's
            if constexpr (SPLIT) {
                f_cur = a_f; th_cur = a_th; inf_pr = a_pr; zmax = a_z; lamc = a_lc;
                wait_slot(MS2_TDONE);
                if (dead) break;
                wave_lds_sync();
                inf_du = uni(res[MS2_DU]); lmax = uni(res[MS2_L]);
                finite = fin_p && uni(res[MS2_FIN]) != 0.0;
            } else read_res();
            cur ^= 1;                                   // ... and the trial pass left the point itself in the other set
            MS2_T1(5);
#ifdef PDP_MS_TIMING
            if (iter_log && it < op.log_rows && lane == 0) {
                // first-chunk wait | Riccati steps | later chunk waits | forward steps | dlam tail | line search + update | sweep prologues (loop top, gather / store maps); total of the iteration
                double* row = iter_log + ((int64_t)b * op.log_rows + it) * 8;
                const long long now_ = __builtin_readcyclecounter();
                for (int k_ = 0; k_ < 7; ++k_) { row[k_] = (double)tm[k_]; tm[k_] = 0; }
                row[7] = (double)(now_ - tmi);
                tmi = now_;
                if (it + op.log_rows / 2 < op.log_rows) { double* r2 = row + (int64_t)(op.log_rows / 2) * 8; for (int k_ = 0; k_ < 8; ++k_) r2[k_] = res[12 + k_]; }
            }
#endif
            gains_ok = false;
            dw = 0.0;
            ++it;
        }
#ifdef PDP_MS_TIMING
        const long long tk3 = __builtin_readcyclecounter();
#endif
        issue(MS2_CMD_EXIT, 0.0, cur, cur);
        if (dead) st |= PDP_MS_INTERNAL;
        {                                               // the iterate, stage-minor in the workspace, into the API arrays
            __threadfence_block();
            const double* sc = Pt(cur);
            rows_out(sc, T + 1, TagNX{}, xb, scratch);
            if (dead) {                                 // (the evaluator writes the controls and the multipliers on its way out - unless it is not there any more)
                rows_out(sc + OU, T, TagNU{}, ub, scratch);
                rows_out(sc + OL, T, TagNX{}, lb, scratch);
            }
        }
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
                go[q] = !have ? 0.0 : (r < NX * NU ? gw[t * GSZ + L::gK(r % NU, r / NU)] : gw[t * GSZ + L::gk(r - NX * NU)]);
            }
            if (!have) st |= PDP_MS_NOGAINS;
        }
        if (lane == 0 && status) status[b] = st;
#ifdef PDP_MS_TIMING
        if (iter_log && op.log_rows >= 4 && it < op.log_rows / 2 - 1 && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const long long tk4 = __builtin_readcyclecounter();
            double* row = iter_log + ((int64_t)b * op.log_rows + op.log_rows / 2 - 1) * 8;
            row[0] = (double)(tk1 - tk0); row[1] = (double)(tk2 - tk1); row[2] = (double)(tk3 - tk2); row[3] = (double)(tk4 - tk3); row[4] = (double)(tk4 - tk0);
            row[5] = (double)tkp; row[6] = (double)tkw; row[7] = (double)tkc;
        }
#endif
    } else {
        // ============================================== evaluator ==============================================
        __builtin_amdgcn_s_setprio(0);
        bool dead = false;
        int last = 0;
        // (this wave's own copies of the launch constants its row offsets derive from: see sopaque)
        const int TSe = sopaque(TS), OUe = NX * TSe, OLe = (NX + NU) * TSe;
        double* const stpe = sopaque(stp);
        // PDP_MS_PREDICT_GUARD, this wave's part: the previous solution - the caller's arrays as they are - into point set 1 and through the residual pass, while the
        // runner forms the prediction in set 0 (same conditions as the runner's `guard`, short of the size of the correction, which does not exist yet)
        if ((op.flags & PDP_MS_WARM) != 0 && (op.flags & PDP_MS_FROM_CONTROLS) == 0 && (op.flags & PDP_MS_PREDICT) != 0 && (op.flags & PDP_MS_PREDICT_GUARD) != 0 && op.dtheta &&
            (op.predict_record || (op.dxdp && op.dudp))) {
            double* s1 = Pt(1);
            for (int i = lane; i < NX; i += 64) s1[i * TSe] = x0[(int64_t)b * NX + i];
            rows_in(xb, T + 1, TagNX{}, s1, 1, scratch);        // (the LDS scratch is the runner's, who has no use for it before its first sweep)
            rows_in(ub, T, TagNU{}, s1 + OUe, 0, scratch);
            rows_in(lb, T, TagNX{}, s1 + OLe, 0, scratch);
            __threadfence_block();
            if (!ms2_wait_ge(ctl + MS2_PARS, 1, ctl)) dead = true;
            trial_pass(PartAll{}, 0.0, 1, 1);
            wave_lds_sync();
            if (lane == 0) { for (int k_ = 0; k_ < 8; ++k_) res[MS2_G0 + k_] = res[MS2_F + k_]; }
            __threadfence_block();
            f3_signal(ctl + MS2_GDONE, 1);
        }
#ifdef PDP_MS_TIMING      // cumulative cycles of the evaluator: trial passes | updates | backward chunk evaluations | forward chunk evaluations | dlam | terminal | waits inside a sweep | number of trial passes
        long long et[8] = {0, 0, 0, 0, 0, 0, 0, 0}, et0 = 0;
#define MS2_E0() et0 = __builtin_readcyclecounter()
#define MS2_E1(k) do { const long long now_ = __builtin_readcyclecounter(); et[k] += now_ - et0; et0 = now_; } while (0)
#else
#define MS2_E0()
#define MS2_E1(k)
#endif
        // Restoration (see the runner's line search): x_{t+1} = f(x_t, u_t) from the fixed x_0 in point set `cur`, multipliers zeroed.  The one serial pass of this
        // solver - every lane computes the same recursion, the controls of 64 stages are fetched by one coalesced load per component and broadcast per step.
        auto restore = [&](int cur) {
            PDP_MS2_PAR();
            double* __restrict__ ps = Pt(cur);
            double xr[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xr[i] = uni(ps[i * TSe]);
            for (int base = 0; base < T; base += 64) {
                const int tl_ = base + lane < T ? base + lane : T - 1;
                double ul[NU];
#pragma unroll
                for (int i = 0; i < NU; ++i) ul[i] = ps[OUe + i * TSe + tl_];
                const int cnt = T - base < 64 ? T - base : 64;
                for (int s_ = 0; s_ < cnt; ++s_) {
                    double uc[NU], v[NX];
#pragma unroll
                    for (int i = 0; i < NU; ++i) uc[i] = readlane_f64(ul[i], s_);
                    Mdl::dyn(xr, uc, th, pc, v);
#pragma unroll
                    for (int i = 0; i < NX; ++i) { xr[i] = v[i]; if (lane == 0) ps[i * TSe + base + s_ + 1] = v[i]; }
                }
            }
            for (int q = lane; q < NX * TSe; q += 64) ps[OLe + q] = 0.0;
            __threadfence_block();
        };
        // multiplier step of the stages [t0, t0 + cnt): dlam_t = P_{t+1} dx_{t+1} + W_{t+1}   (PDP.py:604), lane = stage.  `pp`: the stage's (P, W) record,
        // `d`: dx_{t+1}.  Small systems: the full matrix; else the upper triangle row by row, in batches of rows whose loads are all requested before the
        // first FMA that needs one; element (j, k), k >= j, serves acc[j] and, off the diagonal, acc[k].
        auto dlam_row = [&](const double* pp, const double (&d)[NX], double (&acc)[NX]) {
#pragma unroll
            for (int i = 0; i < NX; ++i) acc[i] = pp[L::pW(i)];
            if constexpr (SMALL) {
                double pv[NX * NX];
#pragma unroll
                for (int k = 0; k < NX * NX; ++k) pv[k] = pp[k];
#pragma unroll
                for (int j = 0; j < NX; ++j)
#pragma unroll
                    for (int i = 0; i < NX; ++i) acc[i] = fma(pv[i * NX + j], d[j], acc[i]);
            } else {
                constexpr int RB_ = 4;
#pragma unroll
                for (int j0 = 0; j0 < NX; j0 += RB_) {
                    double pv[RB_][NX];
#pragma unroll
                    for (int jj = 0; jj < RB_; ++jj)
#pragma unroll
                        for (int k = 0; k < NX; ++k) pv[jj][k] = (j0 + jj < NX && k >= j0 + jj) ? pp[L::pk(j0 + jj < NX ? j0 + jj : 0, k)] : 0.0;
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int jj = 0; jj < RB_; ++jj)
#pragma unroll
                        for (int k = 0; k < NX; ++k)
                            if (j0 + jj < NX && k >= j0 + jj) {
                                const int j = j0 + jj;
                                acc[j] = fma(pv[jj][k], d[k], acc[j]);
                                if (k != j) acc[k] = fma(pv[jj][k], d[j], acc[k]);
                            }
                }
            }
        };
        // route 1 (batches that leave the memory system idle): every lane reads its stage's record straight from the workspace
        auto dlam_chunk = [&](int t0, int cnt) {
            if (lane < cnt) {
                const int t = t0 + lane;
                const unsigned o8 = 8u * (unsigned)t;
                double d[NX], acc[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) d[i] = sm_ld(stpe + i * TSe, o8 + 8u);
                dlam_row(pw + (int64_t)t * PWSZ, d, acc);
#pragma unroll
                for (int i = 0; i < NX; ++i) sm_st(stpe + OLe + i * TSe, o8, acc[i]);
            }
        };
        // route 2 (four trajectories per CU): the records of a block of stages are contiguous in the workspace - they are copied into a free pool buffer
        // with coalesced loads (64 consecutive doubles per wave instruction) and read from LDS by the lane of their stage.  Straight from the workspace every
        // lane sits in its own 840-byte record: 25 lanes x 105 loads = 2600 cache-line requests per chunk and trajectory, which at four trajectories per
        // CU kept the address unit busy for ~17 k cycles per chunk (profiles/r03_ms2_phase_timing_v2.txt) - and slowed the runner's own loads beside it.
        auto dlam_staged = [&](int t0, int cnt, double* sbuf, auto cap_tag) {
            constexpr int CAP = decltype(cap_tag)::value;             // doubles of LDS at sbuf
            constexpr int PSTR = PWSZ | 1;                            // odd record stride in LDS (bank spread over the lanes of a block)
            constexpr int NBMAX = CAP / (PSTR + NX) < 64 ? CAP / (PSTR + NX) : 64;
            constexpr int NQ = (NBMAX * PWSZ + 63) / 64, NQD = (NBMAX * NX + 63) / 64;      // loads per lane: EVERY load of a block is in flight before the first LDS store
            const int nblk = (cnt + NBMAX - 1) / NBMAX, nb0 = (cnt + nblk - 1) / nblk;
            double* dl = sbuf + NBMAX * PSTR;
            for (int s0 = 0; s0 < cnt; s0 += nb0) {
                const int nb = min(nb0, cnt - s0), n = nb * PWSZ, nd = nb * NX;
                const double* src = pw + (int64_t)(t0 + s0) * PWSZ;
                wave_lds_sync();
                double v[NQ], w[NQD];
#pragma unroll
                for (int q = 0; q < NQ; ++q) { const int idx = lane + 64 * q; v[q] = src[idx < n ? idx : 0]; }
#pragma unroll
                for (int q = 0; q < NQD; ++q) { const int k = lane + 64 * q, kk = k < nd ? k : 0, j = kk / nb, sidx = kk - j * nb; w[q] = stpe[j * TSe + t0 + s0 + sidx + 1]; }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int idx = lane + 64 * q;
                    if (idx < n) { if constexpr (PSTR == PWSZ) sbuf[idx] = v[q]; else sbuf[(idx / PWSZ) * PSTR + idx % PWSZ] = v[q]; }
                }
#pragma unroll
                for (int q = 0; q < NQD; ++q) { const int k = lane + 64 * q; if (k < nd) { const int j = k / nb, sidx = k - j * nb; dl[sidx * NX + j] = w[q]; } }
                wave_lds_sync();
                if (lane < nb) {
                    const unsigned o8 = 8u * (unsigned)(t0 + s0 + lane);
                    double d[NX], acc[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) d[i] = dl[lane * NX + i];
                    dlam_row(sbuf + lane * PSTR, d, acc);
#pragma unroll
                    for (int i = 0; i < NX; ++i) sm_st(stpe + OLe + i * TSe, o8, acc[i]);
                }
            }
            wave_lds_sync();
        };
        using Cap1 = std::integral_constant<int, L::BUF>;
        using Cap2 = std::integral_constant<int, 2 * L::BUF>;
        for (;;) {
            if (!ms2_wait_ge(ctl + MS2_SEQ, last + 1, ctl)) break;
            last = ms2_load(ctl + MS2_SEQ);
            wave_lds_sync();
            const int type = uni(ctl[MS2_TYPE]), cur = uni(ctl[MS2_CUR]), dst = uni(ctl[MS2_DST]), csrc = uni(ctl[MS2_CSRC]);
            const double alpha = uni(res[MS2_ALPHA]);
            if (type == MS2_CMD_EXIT) {                 // its share of the result: u and lambda of the final iterate into the API arrays (the runner writes x)
                const double* sc = Pt(cur);
                rows_out(sc + OUe, T, TagNU{}, ub, pool);
                rows_out(sc + OLe, T, TagNX{}, lb, pool);
                break;
            }
            MS2_E0();
            if (type == MS2_CMD_RESTORE) restore(cur);
            if (type == MS2_CMD_TRIAL || type == MS2_CMD_TRIAL_SWEEP || type == MS2_CMD_RESTORE) {
                const bool shared = SPLIT && type == MS2_CMD_TRIAL_SWEEP && dst != cur;      // (the starting point's TRIAL_SWEEP has dst == cur: all of it here)
                if (shared) trial_pass(PartDual{}, alpha, cur, dst);       // (the line search: the runner does the primal half meanwhile)
                else trial_pass(PartAll{}, alpha, cur, dst);
                __threadfence_block();
                MS2_E1(0);
#ifdef PDP_MS_TIMING
                et[7] += 1;
                if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) res[12 + k_] = (double)et[k_];
#endif
                f3_signal(ctl + MS2_TDONE, last);
                // the sweep below reads the trial (x, u) and the defects the runner's half of the pass writes
                if (shared && !ms2_wait_ge(ctl + MS2_PDONE, last, ctl)) break;
            }
            if (type == MS2_CMD_SWEEP || type == MS2_CMD_TRIAL_SWEEP) {
                const int sw = type == MS2_CMD_SWEEP ? cur : dst;          // the set the sweep linearises at
                const double* __restrict__ ps = Pt(sw);
                const double* __restrict__ rd = Rs(sw);
                const double* __restrict__ cdef = (type == MS2_CMD_SWEEP && csrc) ? csoc : rd + OLe;       // defects of the point, or the constraint block of a second-order correction
                bool aborted = false;
                auto stop = [&]() { aborted = aborted || ms2_load(ctl + MS2_ABORT) == last; return aborted || dead; };
                // terminal stage: hxx(x_T) entries and the terminal gradient
                if (lane == 0) fin[0] = 0.0;
                for (int i = lane; i < Mdl::FIN_NCONST; i += 64) fin[1 + i] = Mdl::fin_const(i);
                if (lane == 0) {
                    PDP_MS2_PAR();
                    double xT[NX];
#pragma unroll
                    for (int i = 0; i < NX; ++i) { xT[i] = ps[i * TSe + T]; dlT[i] = rd[i * TSe + T]; }      // (one lane: plain indexing)
                    PackedSink s{fin + L::NCFIN};
                    Mdl::eval_fin(xT, nullptr, nullptr, th, pc, s);
                }
                MS2_E1(5);
                // backward chunks: lane = stage evaluates F, G, Hxx, Hxu, Huu at (x_t, u_t, lambda_{t+1}); defect and Lagrangian gradients come from the residual set
                for (int g = 0; g < nchunk && !stop(); ++g) {
                    int t0, cnt;
                    bchunk(g, t0, cnt);
                    if (g >= 2) {
                        bool freed = false;
                        int n = 0;
                        while (!(freed = ms2_load(ctl + MS2_CONS) >= g - 1) && !stop()) { __builtin_amdgcn_s_sleep(PDP_MS2_SLEEP); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!freed) break;
                    }
                    MS2_E1(6);
                    if (lane < cnt) {
                        PDP_MS2_PAR();
                        const int t = t0 + lane;
                        const unsigned o8 = 8u * (unsigned)t;
                        double xc[NX], uc[NU], lc[NX];
                        double* row = pool + (g & 1) * L::BUF + lane * BS;
                        // (every load of the stage first, then the LDS stores: interleaved, each store waits for its loads - a trip to memory per component)
                        double cc[NX], gx[NX], gu[NU];
#pragma unroll
                        for (int i = 0; i < NX; ++i) { xc[i] = sm_ld(ps + i * TSe, o8); lc[i] = sm_ld(ps + OLe + i * TSe, o8); cc[i] = sm_ld(cdef + i * TSe, o8); gx[i] = sm_ld(rd + i * TSe, o8); }
#pragma unroll
                        for (int i = 0; i < NU; ++i) { uc[i] = sm_ld(ps + OUe + i * TSe, o8); gu[i] = sm_ld(rd + OUe + i * TSe, o8); }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < NX; ++i) { row[L::C0 + i] = cc[i]; row[L::RX + i] = gx[i]; }
#pragma unroll
                        for (int i = 0; i < NU; ++i) row[L::RU + i] = gu[i];
                        PackedSink s{row};
                        Mdl::eval_sol(xc, uc, lc, th, pc, s);
                        row[L::CB0] = 0.0;
                        row[L::ONEB] = 1.0;
#pragma unroll
                        for (int i = 0; i < Mdl::SOL_NCONST; ++i) row[L::CB0 + 1 + i] = Mdl::sol_const(i);
                    }
                    f3_signal(ctl + MS2_PROD, g + 1);
                    MS2_E1(2);
                }
                // forward chunks: F', G', the defect and the gradients the directional derivative needs; behind each consumed chunk the multiplier step
                for (int c = 0; c < nchunkF && !stop(); ++c) {
                    const int g = nchunk + c;
                    int t0, cnt;
                    fchunk(c, t0, cnt);
                    if (g >= 2) {
                        bool freed = false;
                        int n = 0;
                        while (!(freed = ms2_load(ctl + MS2_CONS) >= g - 1) && !stop()) { __builtin_amdgcn_s_sleep(PDP_MS2_SLEEP); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!freed) break;
                    }
                    MS2_E1(6);
                    if (lane < cnt) {
                        PDP_MS2_PAR();
                        const int t = t0 + lane;
                        const unsigned o8 = 8u * (unsigned)t;
                        double xc[NX], uc[NU];
                        double* row = pool + (g & 1) * L::BUF + lane * FS;
                        double cc[NX], gx[NX], gu[NU];
#pragma unroll
                        for (int i = 0; i < NX; ++i) { xc[i] = sm_ld(ps + i * TSe, o8); cc[i] = sm_ld(cdef + i * TSe, o8); gx[i] = sm_ld(rd + i * TSe, o8 + 8u); }
#pragma unroll
                        for (int i = 0; i < NU; ++i) { uc[i] = sm_ld(ps + OUe + i * TSe, o8); gu[i] = sm_ld(rd + OUe + i * TSe, o8); }
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < NX; ++i) { row[L::FC0 + i] = cc[i]; row[L::FRX + i] = gx[i]; }
#pragma unroll
                        for (int i = 0; i < NU; ++i) row[L::FRU + i] = gu[i];
                        PackedSink s{row};
                        Mdl::eval_solf(xc, uc, nullptr, th, pc, s);
                        row[L::CF0] = 0.0;
                        row[L::ONEF] = 1.0;
#pragma unroll
                        for (int i = 0; i < Mdl::SOLF_NCONST; ++i) row[L::CF0 + 1 + i] = Mdl::solf_const(i);
                    }
                    f3_signal(ctl + MS2_PROD, g + 1);
                    MS2_E1(3);
                    if (c >= 1) {       // the runner has left chunk c - 1 (it could not start chunk c before the signal above): its dx are in memory
                        bool got = false;
                        int n = 0;
                        while (!(got = ms2_load(ctl + MS2_CONS) >= g) && !stop()) { __builtin_amdgcn_s_sleep(PDP_MS2_SLEEP); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                        if (!got) break;
                        MS2_E1(6);
                        // (the consumed chunk's buffer is free until chunk c + 1 is evaluated)
                        int tp, cp_;
                        fchunk(c - 1, tp, cp_);
                        if constexpr (AUG) { (void)tp; (void)cp_; }      // (homogeneous form: the runner forms dlam in its forward steps)
                        else if constexpr (TPW == 4 && PDP_MS2_DLAM_STAGED) dlam_staged(tp, cp_, pool + ((g - 1) & 1) * L::BUF, Cap1{});
                        else dlam_chunk(tp, cp_);
                        MS2_E1(4);
                    }
                }
                if (!stop()) {
                    bool got = false;
                    int n = 0;
                    while (!(got = ms2_load(ctl + MS2_CONS) >= nchunk + nchunkF) && !stop()) { __builtin_amdgcn_s_sleep(PDP_MS2_SLEEP); if (++n > (1 << 22)) { f3_signal(ctl + MS2_DEAD, 1); dead = true; } }
                    MS2_E1(6);
                    if (got) {          // the last chunk: both pool buffers are free - its records in ONE block, one trip to memory
                        int tp, cp_;
                        fchunk(nchunkF - 1, tp, cp_);
                        if constexpr (AUG) { (void)tp; (void)cp_; }
                        else if constexpr (TPW == 4 && PDP_MS2_DLAM_STAGED) dlam_staged(tp, cp_, pool, Cap2{});
                        else dlam_chunk(tp, cp_);
                    }
                }
                __threadfence_block();
                MS2_E1(4);
            }
            if (dead) break;
#ifdef PDP_MS_TIMING
            if (lane == 0) for (int k_ = 0; k_ < 8; ++k_) res[12 + k_] = (double)et[k_];
#endif
            f3_signal(ctl + MS2_DONE, last);
        }
    }
#undef PDP_MS2_PAR
#undef MS2_T0
#undef MS2_T1
#undef MS2_E0
#undef MS2_E1
}

template <class Mdl>
__host__ inline int64_t ms2_ws_bytes(int B, int T, int max_iter) { return (int64_t)B * Ms2Layout<Mdl>::ws_doubles(T, max_iter < 0 ? 0 : max_iter) * (int64_t)sizeof(double); }

}  // namespace pdp
