// pdp_fused3_kernels.h - the fused OC unit (forward + costates + aux system in LDS + Riccati + PDP gradient) as a PRODUCER / CONSUMER pair
// of wavefronts per trajectory: oc_pdp_fused3_kernel.  Same inputs, outputs, workspace and arithmetic as oc_pdp_fused_kernel
// (pdp_model_kernels.h); results are bit-identical (the same generated code and the same MFMA sequences run, only on two waves).
//
// Why.  Inside ONE wavefront fp64 MFMA and VALU instructions do not overlap (profiles/r01_probe_mfma_overlap.txt) and a lone wave reaches
// about a third of the SIMD's fp64 VALU rate; two wavefronts on one SIMD do overlap one's MFMAs with the other's VALU / LDS work
// (profiles/r02_probe_two_waves_per_simd.txt).  At the headline batch (1024 trajectories = the 1024 SIMDs of an MI355X) there is no second
// trajectory to put on a SIMD, so the trajectory's own work is split by KIND, not by halves of a step (that was oc_pdp_fused2_kernel:
// two synchronisations per time step, slower):
//     runner    (waves 0..3 of a workgroup):  rollout, then the Riccati backward loops and the forward sensitivity loops - the serial,
//                                              MFMA-heavy chains - over pools of aux-system entries it finds ready in LDS;
//     evaluator (waves 4..7):                  everything that is lane-per-time-step or off the Riccati chain - the generated Jacobian /
//                                              Hessian code (eval_patha / pathb / fwd), the costate recursion, the terminal condition, the
//                                              loss terms - one chunk of time steps AHEAD of the runner, into a double-buffered pool.
// The two meet once per CHUNK (about 13 time steps) at a pair of LDS counters (produced / consumed), not once per step.
//
// Placement.  A workgroup is 512 threads = 4 trajectories: waves w and w + 4 land on the same SIMD (probes/wave_placement_probe.hip: 1024 of
// 1024 pairs, every SIMD hosts exactly two waves when 256 such workgroups are resident), so every SIMD runs one runner and the evaluator
// of the same trajectory, 256 registers each.  (With 128-thread workgroups the two waves of a workgroup go to DIFFERENT SIMDs.)
// Each trajectory owns a 40 KB slice of the workgroup's 160 KB of LDS.
#pragma once
#include "pdp_model_kernels.h"

#ifndef PDP_F3_GAIN_AHEAD
#define PDP_F3_GAIN_AHEAD 3         // forward sweep: the feedback gains of step t + PDP_F3_GAIN_AHEAD are requested during step t (1, 2 or 3; profiles/r04_fused3_attempts.txt)
#endif
#ifndef PDP_F3_ROLLOUT_LANES
#define PDP_F3_ROLLOUT_LANES 0      // 1: the rollout parks x_{t+1} in lane t and stores blocks of 64 steps (see the rollout loop)
#endif

namespace pdp {

#ifdef PDP_PHASE_TIMING
__device__ long long g_rb_stamp[16];      // cycle stamps inside the last backward step of trajectory 0 (PDP_RB_T in pdp_riccati.h)
#endif

template <class Mdl>
struct Fused3Layout {
    using L = FusedLayout<Mdl>;
    static constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP;
    static constexpr int NCB = 1 + Mdl::PATHA_NCONST + Mdl::PATHB_NCONST, NCF = 1 + Mdl::FWD_NCONST, NCFIN = 1 + Mdl::FIN_NCONST;     // [0.0 | constants]
    // backward pool row: [patha | pathb | 0.0 | constants of both groups]; lambda_{t+1} (NX doubles, input of the pathb evaluation) is parked in
    // the first slots of the pathb region, which the same lane overwrites only after it has read them.  EVERY row carries its own zero and
    // its own copy of the (few) constant entries, so that every tile element - varying, constant or absent - sits at the same distance
    // from one row to the next: the time step becomes an immediate offset of the ds_read and the per-lane address registers are updated
    // once per U steps instead of once per step (the one-wave kernel keeps one constant pool and per-lane strides of 0 or one row).
    static constexpr int NA = Mdl::PATHA_NVAR, NB = Mdl::PATHB_NVAR > 16 ? Mdl::PATHB_NVAR : 16;      // (16: the costate tile's column 0 is stored whole)
    static constexpr int CB0 = NA + NB;                                          // slot of the row's 0.0; the constants follow
    static constexpr int BSTRIDE = (CB0 + NCB) | 1;
    static constexpr int CF0 = Mdl::FWD_NVAR + NX + NU;                          // forward row: [fwd | x - x_demo | u - u_demo | 0.0 | constants]
    static constexpr int FSTRIDE = (CF0 + NCF) | 1;
    // offsets (doubles) inside a trajectory's slice
    static constexpr int FIN = RICCATI_SCRATCH;                 // [0.0 | terminal constants | terminal entries]   (behind the Riccati scratch; runner only)
    static constexpr int PAR = FIN + NCFIN + Mdl::FIN_NVAR;      // theta (NP) | theta-only precomputed values (NPC)
    static constexpr int DLT = PAR + NP + Mdl::NPC;              // NX: lambda_T staging (evaluator, start) / x_T - xdemo_T (runner, end)
    static constexpr int MISC = DLT + NX;                        // 8 doubles: ints 0..7 = counters | [4] evaluator's loss sum | [5] dead slot
    static constexpr int POOL = MISC + 8;                        // two buffers of BUF doubles
    static constexpr int SLICE = 160 * 1024 / 8 / 4;
    static constexpr int BUF = (SLICE - POOL) / 2;
    static constexpr int ROWS = BUF / BSTRIDE < 64 ? BUF / BSTRIDE : 64;       // time steps per backward chunk
    static constexpr int ROWSF = BUF / FSTRIDE < 64 ? BUF / FSTRIDE : 64;      // ... per forward chunk (shorter rows)
};

// the kernel applies when the rollout staging fits the pool area and a chunk holds a useful number of steps
template <class Mdl>
__host__ __device__ constexpr bool fused3_ok(int T) {
    using F3 = Fused3Layout<Mdl>;
    return Mdl::NX > 4 && F3::ROWS >= 4 && F3::ROWSF >= 4 && (T + 1) * Mdl::NX + T * Mdl::NU <= 2 * F3::BUF;
}

typedef unsigned f3_u2 __attribute__((ext_vector_type(2)));
typedef unsigned f3_u4 __attribute__((ext_vector_type(4)));

// counters in LDS: release / acquire at workgroup scope (LDS and - for the trajectory the runner leaves in global memory - the CU's L1)
PDP_DEV void f3_signal(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
PDP_DEV void f3_wait_ge(int* f, int v) {
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(2);
}

// Gathers over uniform rows: off[r] = slot (in doubles, inside a row) of tile element (lane, r); absent elements read the row's 0.0
struct Gather3 { int off[4]; };
template <class CodeFn>
PDP_DEV void make_gather3(Gather3& g, int lane, int c0, CodeFn code_of /* (row, col) -> code >= 0, -1 (zero) or <= -2 (constant) */) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int code = code_of(tile_row(lane, r), tile_col(lane));
        g.off[r] = code >= 0 ? code : (code == -1 ? c0 : c0 + 1 + (-2 - code));
    }
}
struct Run3 { unsigned cur[4]; };       // absolute LDS byte addresses of the four elements in the row the run is positioned at
PDP_DEV Run3 run3_at(const Gather3& g, const double* row) {
    Run3 r;
    const unsigned base = lds_addr(row);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.cur[k] = base + 8u * (unsigned)g.off[k];
    return r;
}
template <int NR = 4>
PDP_DEV d4 read3(const Run3& r, unsigned imm) {      // imm: byte distance of the wanted row from the run's row - a literal after unrolling
    d4 v = zero4();
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = *(PDP_LDS const double*)(uintptr_t)(r.cur[k] + imm);
    return v;
}
template <int NR = 4>
PDP_DEV void move3(Run3& r, int bytes) {
#pragma unroll
    for (int k = 0; k < NR; ++k) r.cur[k] += (unsigned)bytes;
}

// tile -> array through range-checked buffer stores: voff[r] = byte offset of element (lane, r) inside one time step's block, or out of range (dropped by the
// hardware, like every lane of a resource of size 0 = an output that was not asked for): no predicated store blocks in the step loops
struct F3StoreMap { unsigned voff[4]; };
PDP_DEV F3StoreMap f3_store_map(int R, int C, int ld, int coff, int lane) {
    F3StoreMap m;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = tile_row(lane, r), col = tile_col(lane) - coff;
        m.voff[r] = (row < R && col >= 0 && col < C) ? 8u * (unsigned)(row * ld + col) : 0x80000000u;
    }
    return m;
}
template <int NR = 4, class RS>
PDP_DEV void f3_bstore(RS rs, unsigned soff, const F3StoreMap& m, const d4 v) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double x = v[r];          // (through a scalar copy: see DESIGN.md section 8, finding 5)
        f3_u2 w;
        w.x = (unsigned)__double2loint(x); w.y = (unsigned)__double2hiint(x);
        __builtin_amdgcn_raw_buffer_store_b64(w, rs, m.voff[r], soff, 0);
    }
}

// TPW trajectories per workgroup of 2 TPW waves (runner = wave j, evaluator = wave j + TPW).  TPW = 4 (512 threads): the pair shares a SIMD - the layout for
// batches that fill the chip (>= 1024 trajectories).  TPW = 2 / 1 (a 512-trajectory shard of C4, small batches): a CU then hosts at most two / one
// trajectory, and the two waves of a trajectory sit on DIFFERENT SIMDs - the evaluator no longer competes with the runner's MFMA chain for issue slots
// (profiles/r03_fused3_small_batch.txt).
// RIC (the instantiation that runs when sensitivity outputs are asked for: dxdp, dudp, and - pdp_oc_pdp_grad_sens_batched - the Riccati record): the runner also
// leaves the Riccati matrices P_{t+1}, W_{t+1} of every stage (PP[t], WW[t] of the reference's lqrSolver, PDP.py:561-580) in `riccati` [B][T][n n + n p + 1] - with
// dxdp / dudp they give the first-order change of the optimal (x, u, lambda) with theta (pdp_oc_predict_batched) - and all three are written with range-checked
// buffer stores (an output that is NULL is a resource of size 0).  A template parameter, not a run-time branch: the default kernel keeps its instruction stream.
template <class Mdl, int TPW = 4, bool RIC = false>
__global__ void __launch_bounds__(128 * TPW) oc_pdp_fused3_kernel(int B, int T, int flags, const double* __restrict__ x0, const double* __restrict__ u,
                                                            const double* __restrict__ theta, int tb, const double* __restrict__ demo_x,
                                                            const double* __restrict__ demo_u, double* __restrict__ x, double* __restrict__ lam,
                                                            double* __restrict__ loss, double* __restrict__ grad, double* __restrict__ dxdp,
                                                            double* __restrict__ dudp, int32_t* __restrict__ status, double* __restrict__ ws_gain,
                                                            double* __restrict__ riccati, float* __restrict__ prec) {
    using F3 = Fused3Layout<Mdl>;
    constexpr int NX = Mdl::NX, NU = Mdl::NU, NP = Mdl::NP, M = NU;
    constexpr int GSZ = fused_gain_doubles<Mdl>(), GSZ0 = fused_gain0_doubles<Mdl>();
    constexpr int BS = F3::BSTRIDE, FS = F3::FSTRIDE;
    constexpr int U = 4;                                    // time steps per address update in the runner's loops
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    // the wave index is uniform over the wave - said explicitly, or every pointer derived from it (trajectory, workspace, LDS slice) would be
    // carried per lane and every global access would pay 64-bit VALU address arithmetic
    static_assert(TPW == 1 || TPW == 2 || TPW == 4, "trajectories per workgroup");
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, slot = wid & (TPW - 1);
    const bool runner = wid < TPW;
    const int b = blockIdx.x * TPW + slot;
    double* lds = lds_all + slot * F3::SLICE;
    double* scratch = lds;
    double* fin = lds + F3::FIN;
    double* par = lds + F3::PAR;
    double* dlT = lds + F3::DLT;
    double* misc = lds + F3::MISC;
    int* fl = (int*)misc;                                   // [0] stage (1: parameters in LDS, 2: trajectory in memory) [1] unused
                                                            // [2] chunks produced [3] chunks consumed [4] evaluator finished
    double* pool = lds + F3::POOL;
    if (runner && lane < 8) fl[lane] = 0;
    __syncthreads();                                        // the only workgroup barrier: counters zeroed before anyone polls them
    if (b >= B) return;
    const d4 z = zero4();
#define PDP_F3_PAR()                                                \
    double th[NP], pc[Mdl::NPC];                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) th[i_] = par[i_]; \
    _Pragma("unroll") for (int i_ = 0; i_ < Mdl::NPC; ++i_) pc[i_] = par[NP + i_]
    double* xb = x + (int64_t)b * (T + 1) * NX;
    double* lb = lam + (int64_t)b * T * NX;
    const double* ub = u + (int64_t)b * T * NU;
    const double* dxb = demo_x + (int64_t)b * (T + 1) * NX;
    const double* dub = demo_u + (int64_t)b * T * NU;
    double* gw = ws_gain + (int64_t)b * T * GSZ;
    const bool given = (flags & PDP_OC_GIVEN_TRAJ) != 0;
    // backward chunks, last time steps first, of equal length.  (A short first chunk - the runner idles until the first chunk is ready - was
    // tried twice: the evaluator, at low priority, then cannot fill the SECOND chunk within the few steps the runner spends on the first, and
    // the runner waits there instead: first chunk of 5 steps 0.1107 against 0.1093 ms at the time, 8 | 16 | 16 | 10 steps 210 k against 207 k cycles.)
    const int nchunk = (T + F3::ROWS - 1) / F3::ROWS;
    const int ch = (T + nchunk - 1) / nchunk;
    auto bchunk = [&](int g, int& t0, int& cnt) { const int c = nchunk - 1 - g; t0 = c * ch; cnt = min(ch, T - t0); };      // chunk g covers [t0, t0 + cnt)
    const int nchunkF = (T + F3::ROWSF - 1) / F3::ROWSF;
    const int chF = (T + nchunkF - 1) / nchunkF;            // forward chunks likewise
#ifdef PDP_PHASE_TIMING     // debug builds (probes/phase_timing3.py): cycle stamps of trajectory 0 behind loss[B]
    long long ts[12]; int nts = 0; long long twait = 0, tw0 = 0;
    const long long rt0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 12; ++i) ts[i] = 0;
#define F3_STAMP() ts[nts++] = __builtin_readcyclecounter()
#define F3_W0() tw0 = __builtin_readcyclecounter()
#define F3_W1() twait += __builtin_readcyclecounter() - tw0
#else
#define F3_STAMP()
#define F3_W0()
#define F3_W1()
#endif
    F3_STAMP();

    if (runner) {
        // =========================================== runner ===========================================
        __builtin_amdgcn_s_setprio(3);
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
            f3_signal(fl + 0, 1);
        }
        // ---- rollout x+ = f(x, u, theta): scalar recursion executed uniformly by the wave.  u is staged in the (still unused) pool and read one
        // step ahead; x_{t+1} goes straight to the API output from lane 0 - global stores are counted by vmcnt, which nothing in the loop
        // waits for, while LDS stores would sit in front of the next step's u reads in the in-order LDS counter (probes/rollout_probe.hip)
        double xTr[NX];                                          // x_T stays in registers for the terminal condition (no round trip through memory)
        if (given) {
#pragma unroll
            for (int i = 0; i < NX; ++i) xTr[i] = xb[T * NX + i];
        } else {
            double* us = pool;                                   // T x NU
            for (int i = lane; i < T * NU; i += 64) us[i] = ub[i];
            PDP_F3_PAR();
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
            double un[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) un[i] = us[i];
#if PDP_F3_ROLLOUT_LANES
            // x_{t+1} is parked in LANE (t mod 64) of a register set (2 NX v_cndmask_b32 per step) and blocks of 64 steps go to the API output with NX / 2 stores of
            // full lanes - against NX / 2 stores PER STEP that carry one live lane each and still occupy the address / data path for ~28 cycles (profiles/r04_fused3_attempts.txt)
            double xs[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) xs[i] = 0.0;
            for (int tb_ = 0; tb_ < T; tb_ += 64) {
                const int nb = min(64, T - tb_);
                for (int tl_ = 0; tl_ < nb; ++tl_) {
                    const int t = tb_ + tl_, tn = t + 1 < T ? t + 1 : t;
#pragma unroll
                    for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
                    Mdl::dyn(xc, uc, th, pc, xn);
                    const bool mine = lane == tl_;
#pragma unroll
                    for (int i = 0; i < NX; ++i) { xc[i] = xn[i]; xs[i] = mine ? xn[i] : xs[i]; }
                }
                const unsigned vo = lane < nb ? (unsigned)(lane * NX) * 8u : 0x80000000u;
                const unsigned so = (unsigned)((tb_ + 1) * NX) * 8u;
#pragma unroll
                for (int i = 0; i + 1 < NX; i += 2) {
                    f3_u4 w;
                    w.x = (unsigned)__double2loint(xs[i]); w.y = (unsigned)__double2hiint(xs[i]);
                    w.z = (unsigned)__double2loint(xs[i + 1]); w.w = (unsigned)__double2hiint(xs[i + 1]);
                    __builtin_amdgcn_raw_buffer_store_b128(w, rsX, vo + 8u * i, so, 0);
                }
                if constexpr (NX & 1) {
                    f3_u2 w;
                    w.x = (unsigned)__double2loint(xs[NX - 1]); w.y = (unsigned)__double2hiint(xs[NX - 1]);
                    __builtin_amdgcn_raw_buffer_store_b64(w, rsX, vo + 8u * (NX - 1), so, 0);
                }
            }
#else
            for (int t = 0; t < T; ++t) {
                const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
                for (int i = 0; i < NU; ++i) { uc[i] = un[i]; un[i] = us[tn * NU + i]; }
                Mdl::dyn(xc, uc, th, pc, xn);
#pragma unroll
                for (int i = 0; i < NX; ++i) xc[i] = xn[i];
                // x_{t+1} to the API output from lane 0, without a branch: a buffer store whose offset is out of range in every other lane
                {
                    const unsigned so = (unsigned)((t + 1) * NX) * 8u;
#pragma unroll
                    for (int i = 0; i + 1 < NX; i += 2) {
                        f3_u4 w;
                        w.x = (unsigned)__double2loint(xn[i]); w.y = (unsigned)__double2hiint(xn[i]);
                        w.z = (unsigned)__double2loint(xn[i + 1]); w.w = (unsigned)__double2hiint(xn[i + 1]);
                        __builtin_amdgcn_raw_buffer_store_b128(w, rsX, xvoff + 8u * i, so, 0);
                    }
                    if constexpr (NX & 1) {
                        f3_u2 w;
                        w.x = (unsigned)__double2loint(xn[NX - 1]); w.y = (unsigned)__double2hiint(xn[NX - 1]);
                        __builtin_amdgcn_raw_buffer_store_b64(w, rsX, xvoff + 8u * (NX - 1), so, 0);
                    }
                }
            }
#endif
#pragma unroll
            for (int i = 0; i < NX; ++i) xTr[i] = xc[i];
        }
        f3_signal(fl + 0, 2);                                    // release: the trajectory is in memory, the staging area is free
        F3_STAMP();

        // ---- terminal condition: P = hxx(x_T), W = hxe(x_T) - evaluated here, while the evaluator fills the first chunk
        bool ok = true;
        d4 P, W2;
        {
            if (lane == 0) fin[0] = 0.0;
            for (int i_ = lane; i_ < Mdl::FIN_NCONST; i_ += 64) fin[1 + i_] = Mdl::fin_const(i_);
            if (lane == 0) {
                PDP_F3_PAR();
                PackedSink s{fin + F3::NCFIN};
                Mdl::eval_fin(xTr, nullptr, nullptr, th, pc, s);
            }
            wave_lds_sync();
            Gather gP, gW;
            make_gather(gP, lane, F3::NCFIN, 0, [](int r, int c) { return (r < NX && c < NX) ? Mdl::fin_code(0, r * NX + c) : -1; });
            make_gather(gW, lane, F3::NCFIN, 0, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fin_code(1, r * NP + (c - M)) : -1; });
            P = gather_tile(fin, gP, 0);
            W2 = gather_tile(fin, gW, 0);
        }
        F3_STAMP();

        // ---- backward sweep: Riccati steps over the chunks the evaluator has filled
        {
            constexpr int NA = F3::NA, NCA = Mdl::PATHA_NCONST;
            auto codeA = [](int mat, int i) { return Mdl::patha_code(mat, i); };
            auto codeB = [](int mat, int i) { int c = Mdl::pathb_code(mat, i); return c >= 0 ? c + NA : (c == -1 ? -1 : c - NCA); };
            Gather3 gF, gY, gHxx, gHX, gHU, gGr, gHux;
            make_gather3(gF, lane, F3::CB0, [&](int r, int c) { return (r < NX && c < NX) ? codeA(0, r * NX + c) : -1; });
            make_gather3(gY, lane, F3::CB0, [&](int r, int c) {
                return r >= NX ? -1 : (c < M ? codeA(1, r * NU + c) : (c < M + NP ? codeA(2, r * NP + (c - M)) : -1)); });
            make_gather3(gGr, lane, F3::CB0, [&](int r, int c) { return (r < NX && (c & 3) < NU) ? codeA(1, r * NU + (c & 3)) : -1; });
            make_gather3(gHux, lane, F3::CB0, [&](int r, int c) { return (r < M && c < NX) ? codeB(1, c * NU + r) : -1; });
            make_gather3(gHxx, lane, F3::CB0, [&](int r, int c) { return (r < NX && c < NX) ? codeB(0, r * NX + c) : -1; });
            make_gather3(gHX, lane, F3::CB0, [&](int r, int c) {
                return r >= NX ? -1 : (c < M ? codeB(1, r * NU + c) : (c < M + NP ? codeB(2, r * NP + (c - M)) : -1)); });
            make_gather3(gHU, lane, F3::CB0, [&](int r, int c) {
                return r >= M ? -1 : (c < M ? codeB(3, r * NU + c) : (c < M + NP ? codeB(4, r * NP + (c - M)) : -1)); });
            // gains of a step in the workspace: K [NU x NX] (rows 0..3 of its tile: one register), k [NU x NP], zero sink
            const TileMapBytes mK = make_tile_map_sink(NU, NX, NX, 0, 0, lane, GSZ0 - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
            // Riccati record of a stage (RIC): P_{t+1} [NX x NX] | W_{t+1} [NX x NP] | zero sink
            constexpr int RSZ = oc_riccati_doubles<Mdl>();
            [[maybe_unused]] const F3StoreMap mRP = f3_store_map(NX, NX, NX, 0, lane), mRW = f3_store_map(NX, NP, NP, M, lane);
            [[maybe_unused]] const auto rsR = __builtin_amdgcn_make_buffer_rsrc((void*)(RIC && riccati ? riccati + (int64_t)b * T * RSZ : ws_gain), 0,
                                                                                 RIC && riccati ? (int)((int64_t)T * RSZ * 8) : 0, 0x00020000);
            // packed fp32 prediction record (PredRec, pdp_model_kernels.h)
            [[maybe_unused]] const PredMaps<Mdl> pm(lane);
            [[maybe_unused]] const bool precPW = RIC && prec && !(flags & PDP_OC_RECORD_PRIMAL);      // the P | W part of the record is wanted
            [[maybe_unused]] const auto rsPR = __builtin_amdgcn_make_buffer_rsrc((void*)(precPW ? (void*)(prec + (int64_t)b * T * PredRec<Mdl>::SIZE) : (void*)ws_gain), 0,
                                                                                  precPW ? (int)((int64_t)T * PredRec<Mdl>::SIZE * 4) : 0, 0x00020000);
            constexpr int RB = 8 * BS;                           // bytes per row
            for (int g = 0; g < nchunk; ++g) {
                int t0, cnt;
                bchunk(g, t0, cnt);
                const double* pb = pool + (g & 1) * F3::BUF;
                F3_W0();
                f3_wait_ge(fl + 2, g + 1);
                F3_W1();
                int tl = cnt - 1;
                // the runs sit one row BELOW the step's row: the step reads at +RB, its one-step-ahead requests at +0
                const double* r0 = pb + (tl - 1) * BS;
                Run3 rF = run3_at(gF, r0), rY = run3_at(gY, r0), rHxx = run3_at(gHxx, r0), rHX = run3_at(gHX, r0), rHU = run3_at(gHU, r0),
                     rGr = run3_at(gGr, r0), rHux = run3_at(gHux, r0);
                auto move_all = [&](int bytes) {
                    move3(rF, bytes); move3(rY, bytes); move3(rHxx, bytes); move3(rHX, bytes); move3<1>(rHU, bytes); move3(rGr, bytes); move3<1>(rHux, bytes);
                };
                // F and [G|E] feed the first MFMAs of a step and are requested one step ahead (two register sets in rotation); the Hessian
                // tiles are accumulator inputs of later MFMAs and are requested at the top of their own step.  Step 0 of a chunk requests
                // nothing ahead (no LDS read outside the buffer).
                d4 Fa = read3(rF, RB), Ya = read3(rY, RB), Fb = z, Yb = z;
                auto bstep = [&](int tl, unsigned imm, const d4 Fc, const d4 Yc, d4& Fn, d4& Yn) {      // imm: distance of row tl from the runs
                    const int t = t0 + tl;
#ifdef PDP_PHASE_TIMING_FINE
                    if (blockIdx.x == 0 && threadIdx.x == 0) g_rb_stamp[9] = __builtin_readcyclecounter();
#endif
                    d4 Hxx = read3(rHxx, imm), HX2 = read3(rHX, imm), HU2 = read3<1>(rHU, imm), Grep = read3(rGr, imm), Hux = read3<1>(rHux, imm);
                    if (tl > 0) { Fn = read3(rF, imm - RB); Yn = read3(rY, imm - RB); }
                    RiccatiGains gn;
                    d4 P_old;
                    if constexpr (RIC) {          // (uniform branches: a store to an absent output would be dropped by its size-0 resource, but still issued - 8 to 13 per step)
                        if (riccati) { f3_bstore(rsR, (unsigned)(t * RSZ) * 8u, mRP, P); f3_bstore(rsR, (unsigned)(t * RSZ + NX * NX) * 8u, mRW, W2); }
                        if (precPW) { pred_store(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.P, P); pred_store(rsPR, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pm.W, W2); }
                    }
                    ok = riccati_backward<M, false>(P, W2, Fc, Yc, Grep, Hxx, HX2, HU2, Hux[0], scratch, lane, NP, gn, P_old) && ok;
                    store_all<1>(gw + t * GSZ, mK, gn.K);
                    store_all<1>(gw + t * GSZ + NX * NU, mIK, gn.IK);
#ifdef PDP_PHASE_TIMING_FINE
                    if (blockIdx.x == 0 && threadIdx.x == 0) { g_rb_stamp[11] = g_rb_stamp[10]; g_rb_stamp[10] = __builtin_readcyclecounter(); }
#endif
                };
                // single steps until a whole number of groups of U remains (the register sets move up by copies there)
                for (; (tl + 1) % U != 0; --tl) { bstep(tl, RB, Fa, Ya, Fb, Yb); Fa = Fb; Ya = Yb; move_all(-RB); }
                // groups of U steps: the runs sit on row tl - U, step tl - j reads at (U - j) rows - literal offsets, one address update per group
                move_all(-(U - 1) * RB);
                for (; tl >= U - 1; tl -= U) {
                    static_assert(U == 4, "the group is written out");
                    bstep(tl, (unsigned)(4 * RB), Fa, Ya, Fb, Yb);
                    bstep(tl - 1, (unsigned)(3 * RB), Fb, Yb, Fa, Ya);
                    bstep(tl - 2, (unsigned)(2 * RB), Fa, Ya, Fb, Yb);
                    bstep(tl - 3, (unsigned)(1 * RB), Fb, Yb, Fa, Ya);
                    move_all(-U * RB);
                }
                f3_signal(fl + 3, g + 1);
            }
        }
        bool finite = tile_finite(P) && tile_finite(W2);
        F3_STAMP();

        // ---- forward sweep: sensitivities X_t = dx_t/dtheta, U_t, gradient
        double acc = 0.0, lsum = 0.0;
        const double dT = lane < NX ? xb[T * NX + lane] - dxb[T * NX + lane] : 0.0;      // terminal residual, requested ahead of the loops that hide its latency
        d4 X2 = z;
        {
            constexpr int DLX = Mdl::FWD_NVAR, DLU = Mdl::FWD_NVAR + NX;      // pool slots of x - x_demo, u - u_demo
            Gather3 gFT, gGT, gE, gDX, gDU;
            make_gather3(gFT, lane, F3::CF0, [](int r, int c) { return (r < NX && c < NX) ? Mdl::fwd_code(0, c * NX + r) : -1; });
            make_gather3(gGT, lane, F3::CF0, [](int r, int c) { return (r < M && c < NX) ? Mdl::fwd_code(1, c * NU + r) : -1; });
            make_gather3(gE, lane, F3::CF0, [](int r, int c) { return (r < NX && c >= M && c < M + NP) ? Mdl::fwd_code(2, r * NP + (c - M)) : -1; });
            make_gather3(gDX, lane, F3::CF0, [](int r, int c) { return (r < NX) ? DLX + r : -1; });
            make_gather3(gDU, lane, F3::CF0, [](int r, int c) { return (r < M) ? DLU + r : -1; });
            // feedback gains of step t are fetched one step ahead (each lane re-reads exactly what it stored); K is read back transposed and
            // replicated in the four column blocks (operand form of the 4-row product U = -K X - k)
            const TileMapBytes mKT = to_bytes_sink(make_rep4_map_transposed(NX, NU, NX, lane), GSZ0 - 1), mIK = make_tile_map_sink(NU, NP, NP, 0, M, lane, NU * NP);
#if PDP_F3_GAIN_AHEAD >= 2
            // gains requested TWO or THREE steps ahead: four register sets, set s = gains of step (group base + s); sets 0 .. AHEAD - 1 are loaded when a group starts
            d4 KTs[4], ks[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                if (s_ < PDP_F3_GAIN_AHEAD) {
                    const int ts = s_ < T ? s_ : T - 1;
                    KTs[s_] = -load_all<4>(gw + ts * GSZ, mKT);
                    ks[s_] = -load_all<1>(gw + ts * GSZ + NX * NU, mIK);
                } else { KTs[s_] = z; ks[s_] = z; }
            }
#else
            d4 KTn = -load_all<4>(gw, mKT);
            d4 kn = -load_all<1>(gw + NX * NU, mIK);
#endif
            // sensitivity outputs of the RIC instantiation: buffer stores, an absent output is a resource of size 0
            [[maybe_unused]] const F3StoreMap mSX = f3_store_map(NX, NP, NP, M, lane), mSU = f3_store_map(NU, NP, NP, M, lane);
            [[maybe_unused]] const auto rsSX = __builtin_amdgcn_make_buffer_rsrc((void*)(dxdp ? dxdp + (int64_t)b * (T + 1) * NX * NP : ws_gain), 0,
                                                                                  dxdp ? (int)((int64_t)(T + 1) * NX * NP * 8) : 0, 0x00020000);
            [[maybe_unused]] const auto rsSU = __builtin_amdgcn_make_buffer_rsrc((void*)(dudp ? dudp + (int64_t)b * T * NU * NP : ws_gain), 0,
                                                                                  dudp ? (int)((int64_t)T * NU * NP * 8) : 0, 0x00020000);
            [[maybe_unused]] const PredMaps<Mdl> pmf(lane);
            [[maybe_unused]] const auto rsPRf = __builtin_amdgcn_make_buffer_rsrc((void*)(RIC && prec ? (void*)(prec + (int64_t)b * T * PredRec<Mdl>::SIZE) : (void*)ws_gain), 0,
                                                                                   RIC && prec ? (int)((int64_t)T * PredRec<Mdl>::SIZE * 4) : 0, 0x00020000);
            constexpr int RF = 8 * FS;
            static_assert((U & 1) == 0, "the register sets of the unrolled loops alternate: U must be even");
            for (int c = 0; c < nchunkF; ++c) {
                const int g = nchunk + c, t0 = c * chF, cnt = min(chF, T - t0);
                const double* pb = pool + (g & 1) * F3::BUF;
                F3_W0();
                f3_wait_ge(fl + 2, g + 1);
                F3_W1();
                Run3 rFT = run3_at(gFT, pb), rGT = run3_at(gGT, pb), rE = run3_at(gE, pb), rDX = run3_at(gDX, pb), rDU = run3_at(gDU, pb);
                auto move_all = [&](int bytes) { move3(rFT, bytes); move3<1>(rGT, bytes); move3(rE, bytes); move3(rDX, bytes); move3<1>(rDU, bytes); };
                auto fstep = [&](int tl, unsigned imm, const d4 Xc, d4& Xn, const d4 KTc, const d4 kc, d4& KTnx, d4& knx) {
                    const int t = t0 + tl, tnx = (t + PDP_F3_GAIN_AHEAD < T) ? t + PDP_F3_GAIN_AHEAD : T - 1;
                    KTnx = -load_all<4>(gw + tnx * GSZ, mKT);
                    knx = -load_all<1>(gw + tnx * GSZ + NX * NU, mIK);
                    d4 FT = read3(rFT, imm);
                    d4 GT = read3<1>(rGT, imm);
                    d4 E2 = read3(rE, imm);
                    d4 DX = read3(rDX, imm);                    // (x_t - xd_t)[row] broadcast over columns
                    d4 DU = read3<1>(rDU, imm);
                    d4 U2;
                    riccati_forward(KTc, kc, FT, GT, E2, Xc, U2, Xn);
                    acc += DX[0] * Xc[0] + DX[1] * Xc[1] + DX[2] * Xc[2] + DX[3] * Xc[3] + DU[0] * U2[0];
                    if constexpr (RIC) {
                        if (dxdp || dudp) {
                            f3_bstore(rsSX, (unsigned)(t * NX * NP) * 8u, mSX, Xc);
                            f3_bstore<1>(rsSU, (unsigned)(t * NU * NP) * 8u, mSU, U2);
                        }
                        if (prec) {
                            pred_store(rsPRf, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pmf.X, Xn);       // X_{t+1}
                            pred_store<1>(rsPRf, (unsigned)(t * PredRec<Mdl>::SIZE) * 4u, pmf.U, U2);
                        }
                    } else {
                        if (dxdp) store_dense(dxdp + ((int64_t)b * (T + 1) + t) * NX * NP, NX, NP, NP, 0, M, lane, Xc);
                        if (dudp) store_dense(dudp + ((int64_t)b * T + t) * NU * NP, NU, NP, NP, 0, M, lane, U2);
                    }
                };
                // groups of U steps with literal row offsets, the sensitivity tile and the prefetched gains alternating between two register sets
                d4 Xb;
                int tl = 0;
#if PDP_F3_GAIN_AHEAD >= 2
                static_assert(U == 4 && PDP_F3_GAIN_AHEAD <= 3, "four gain sets rotate through a group of four steps");
                constexpr int A = PDP_F3_GAIN_AHEAD;
                for (; tl + U <= cnt; tl += U) {
                    fstep(tl + 0, (unsigned)(0 * RF), X2, Xb, KTs[0], ks[0], KTs[(0 + A) & 3], ks[(0 + A) & 3]);
                    fstep(tl + 1, (unsigned)(1 * RF), Xb, X2, KTs[1], ks[1], KTs[(1 + A) & 3], ks[(1 + A) & 3]);
                    fstep(tl + 2, (unsigned)(2 * RF), X2, Xb, KTs[2], ks[2], KTs[(2 + A) & 3], ks[(2 + A) & 3]);
                    fstep(tl + 3, (unsigned)(3 * RF), Xb, X2, KTs[3], ks[3], KTs[(3 + A) & 3], ks[(3 + A) & 3]);
                    move_all(U * RF);
                }
                for (; tl < cnt; ++tl) {
                    fstep(tl, 0u, X2, Xb, KTs[0], ks[0], KTs[A], ks[A]);
                    X2 = Xb;
#pragma unroll
                    for (int s_ = 0; s_ < A; ++s_) { KTs[s_] = KTs[s_ + 1]; ks[s_] = ks[s_ + 1]; }
                    move_all(RF);
                }
#else
                d4 KTb, kb;
                for (; tl + U <= cnt; tl += U) {
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        if (j & 1) fstep(tl + j, (unsigned)(j * RF), Xb, X2, KTb, kb, KTn, kn);
                        else fstep(tl + j, (unsigned)(j * RF), X2, Xb, KTn, kn, KTb, kb);
                    }
                    move_all(U * RF);
                }
                for (; tl < cnt; ++tl) { fstep(tl, 0u, X2, Xb, KTn, kn, KTb, kb); X2 = Xb; KTn = KTb; kn = kb; move_all(RF); }
#endif
                f3_signal(fl + 3, g + 1);
            }
        }
        F3_STAMP();
        // terminal term (x_T - xd_T)' X_T   (cartpole_PDP.py:74)
        wave_lds_sync();
        if (lane < NX) { dlT[lane] = dT; lsum += dT * dT; }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) { int row = tile_row(lane, r); if (row < NX) acc += dlT[row] * X2[r]; }
        if (dxdp) store_dense(dxdp + ((int64_t)b * (T + 1) + T) * NX * NP, NX, NP, NP, 0, M, lane, X2);
        finite = finite && tile_finite(X2);
        acc = sum_over_rowgroups(acc);
        lsum = wave_sum(lsum);
        F3_W0();
        f3_wait_ge(fl + 4, 1);
        F3_W1();
        lsum += misc[4];                                        // the evaluator's share: sum over t < T of |x - xd|^2 + |u - ud|^2
        // PDP_OC_PACKED: grad is [B][NP + 1] with the loss in the last column - the row the data-parallel iteration all-gathers
        const int gstride = (flags & PDP_OC_PACKED) ? NP + 1 : NP;
        if (lane >= M && lane < M + NP) grad[(int64_t)b * gstride + (lane - M)] = acc;
        if (lane == 0) { loss[b] = lsum; if (flags & PDP_OC_PACKED) grad[(int64_t)b * gstride + NP] = lsum; }
        int st = 0;
        if (!__all(finite)) st |= PDP_STATUS_NONFINITE;
        if (!ok) st |= PDP_STATUS_PIVOT;
        if (lane == 0 && status) status[b] = st;
#ifdef PDP_PHASE_TIMING
        F3_STAMP();
        if (lane == 0 && b == 0) { long long* o = (long long*)(loss + B); for (int i = 0; i < 12; ++i) o[i] = ts[i]; o[12] = twait; }
        if (lane == 0 && b == 0) { long long* o = (long long*)(loss + B) + 32; for (int i = 0; i < 16; ++i) o[i] = g_rb_stamp[i]; }
        if (lane == 0) { long long* o = (long long*)(loss + B) + 64 + 4 * b; o[0] = rt0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = ts[nts - 1] - ts[0]; o[3] = twait; }
#endif
    } else {
        // ========================================== evaluator ==========================================
#ifndef PDP_F3_EVAL_PRIO
#define PDP_F3_EVAL_PRIO 0
#endif
        __builtin_amdgcn_s_setprio(PDP_F3_EVAL_PRIO);
        F3_W0();
        f3_wait_ge(fl + 0, 2);                                   // parameters in LDS, trajectory in memory
        F3_W1();
        F3_STAMP();
        // ---- terminal condition and the constants of both groups
        d4 Lam = z;                                              // costate tile: column 0 holds lambda_{t+1}; lambda_T = h_x(x_T)
        if (!given) {
            PDP_F3_PAR();
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
        F3_STAMP();
        // ---- backward chunks: (A) lane = time step evaluates F, G, E, c_x -> (C) costates through the chunk on MFMA,
        //      lambda_t = c_x + F_t' lambda_{t+1} -> (B) lane = time step evaluates the lambda-weighted Hessians
        {
            constexpr int NA = F3::NA;
            auto codeA = [](int mat, int i) { return Mdl::patha_code(mat, i); };
            Gather3 gF, gCX;
            make_gather3(gF, lane, F3::CB0, [&](int r, int c) { return (r < NX && c < NX) ? codeA(0, r * NX + c) : -1; });
            make_gather3(gCX, lane, F3::CB0, [&](int r, int c) { return (r < NX && c == 0) ? codeA(3, r) : -1; });
            constexpr int RB = 8 * BS;
            for (int g = 0; g < nchunk; ++g) {
                int t0, cnt;
                bchunk(g, t0, cnt);
                const int bo = (g & 1) * F3::BUF;
                double* pb = pool + bo;
                F3_W0();
                f3_wait_ge(fl + 3, g - 1);                       // the buffer's previous chunk has been consumed
                F3_W1();
                if (lane < cnt) {
                    PDP_F3_PAR();
                    const int t = t0 + lane;
                    double xc[NX], uc[NU];
#pragma unroll
                    for (int i = 0; i < NX; ++i) xc[i] = xb[t * NX + i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                    double* row = pb + lane * BS;
                    PackedSink s{row};
                    Mdl::eval_patha(xc, uc, nullptr, th, pc, s);
                    row[F3::CB0] = 0.0;                          // the row's own zero and constants (see Fused3Layout)
#pragma unroll
                    for (int i = 0; i < Mdl::PATHA_NCONST; ++i) row[F3::CB0 + 1 + i] = Mdl::patha_const(i);
#pragma unroll
                    for (int i = 0; i < Mdl::PATHB_NCONST; ++i) row[F3::CB0 + 1 + Mdl::PATHA_NCONST + i] = Mdl::pathb_const(i);
                }
                wave_lds_sync();
                if (!given) {
                    // F_t and c_x,t do not depend on the recursion: they are requested one step ahead of the MFMA chain that needs them (two
                    // register sets in rotation, like the costate tile itself - the chain of a step reads its predecessor's tile until its
                    // last instruction).  Same addressing as the runner's loops: runs one row below the step's row, literal row offsets.
                    int tl = cnt - 1;
                    const double* r0 = pb + (tl - 1) * BS;
                    Run3 cF = run3_at(gF, r0), cC = run3_at(gCX, r0), wL;
#pragma unroll
                    for (int r = 0; r < 4; ++r) wL.cur[r] = lds_addr(r0) + 8u * (unsigned)(NA + tile_row(lane, r));      // lambda_{t+1} -> row tl (column-0 lanes)
                    auto move_all = [&](int bytes) { move3(cF, bytes); move3(cC, bytes); move3(wL, bytes); };
                    d4 Fa = read3(cF, RB), CXa = read3(cC, RB), Fb = z, CXb = z, Lam2 = z;
                    auto cstep = [&](int tl, unsigned imm, const d4 Fc, const d4 CXc, d4& Fn, d4& CXn, const d4 Lin, d4& Lout) {
                        if (tl > 0) { Fn = read3(cF, imm - RB); CXn = read3(cC, imm - RB); }
                        if (tile_col(lane) == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) *(PDP_LDS double*)(uintptr_t)(wL.cur[r] + imm) = Lin[r];
                        }
                        Lout = mma_tn(Fc, Lin, CXc);    // lambda_t = c_x(x_t,u_t) + F_t' lambda_{t+1}
                    };
                    for (; (tl + 1) % U != 0; --tl) { cstep(tl, RB, Fa, CXa, Fb, CXb, Lam, Lam2); Fa = Fb; CXa = CXb; Lam = Lam2; move_all(-RB); }
                    move_all(-(U - 1) * RB);
                    for (; tl >= U - 1; tl -= U) {
#pragma unroll
                        for (int j = 0; j < U; ++j) {
                            if (j & 1) cstep(tl - j, (unsigned)((U - j) * RB), Fb, CXb, Fa, CXa, Lam2, Lam);
                            else cstep(tl - j, (unsigned)((U - j) * RB), Fa, CXa, Fb, CXb, Lam, Lam2);
                        }
                        move_all(-U * RB);
                    }
                    wave_lds_sync();
                }
                if (lane < cnt) {
                    PDP_F3_PAR();
                    const int t = t0 + lane;
                    double xc[NX], uc[NU], lc[NX];
                    double* row = pb + lane * BS;
#pragma unroll
                    for (int i = 0; i < NX; ++i) xc[i] = xb[t * NX + i];
#pragma unroll
                    for (int i = 0; i < NU; ++i) uc[i] = ub[t * NU + i];
                    if (given) {
#pragma unroll
                        for (int i = 0; i < NX; ++i) lc[i] = lb[t * NX + i];
                    } else {
#pragma unroll
                        for (int i = 0; i < NX; ++i) { lc[i] = row[NA + i]; lb[t * NX + i] = lc[i]; }      // costate is an API output
                    }
                    PackedSink s{row + NA};
                    Mdl::eval_pathb(xc, uc, lc, th, pc, s);
                }
                f3_signal(fl + 2, g + 1);
            }
        }
        F3_STAMP();
        // ---- forward chunks: lane = time step evaluates F', G', E and the loss terms x - x_demo, u - u_demo
        double lsum = 0.0;
        {
            constexpr int DLX = Mdl::FWD_NVAR, DLU = Mdl::FWD_NVAR + NX;
            for (int c = 0; c < nchunkF; ++c) {
                const int g = nchunk + c, t0 = c * chF, cnt = min(chF, T - t0), bo = (g & 1) * F3::BUF;
                F3_W0();
                f3_wait_ge(fl + 3, g - 1);
                F3_W1();
                if (lane < cnt) {
                    PDP_F3_PAR();
                    const int t = t0 + lane;
                    double xc[NX], uc[NU];
                    double* row = pool + bo + lane * FS;
#pragma unroll
                    for (int i = 0; i < NX; ++i) { xc[i] = xb[t * NX + i]; double d = xc[i] - dxb[t * NX + i]; row[DLX + i] = d; lsum += d * d; }
#pragma unroll
                    for (int i = 0; i < NU; ++i) { uc[i] = ub[t * NU + i]; double d = uc[i] - dub[t * NU + i]; row[DLU + i] = d; lsum += d * d; }
                    PackedSink s{row};
                    Mdl::eval_fwd(xc, uc, nullptr, th, pc, s);
                    row[F3::CF0] = 0.0;
#pragma unroll
                    for (int i = 0; i < Mdl::FWD_NCONST; ++i) row[F3::CF0 + 1 + i] = Mdl::fwd_const(i);
                }
                f3_signal(fl + 2, g + 1);
            }
        }
        lsum = wave_sum(lsum);
        if (lane == 0) misc[4] = lsum;
        f3_signal(fl + 4, 1);
#ifdef PDP_PHASE_TIMING
        F3_STAMP();
        if (lane == 0 && b == 0) { long long* o = (long long*)(loss + B) + 16; for (int i = 0; i < 12; ++i) o[i] = ts[i]; o[12] = twait; }
#endif
    }
}

}  // namespace pdp
