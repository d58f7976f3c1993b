// pdp_lqr.hip - model-independent batched kernels of libpdp_hip.so (section A of include/pdp_hip.h):
//   pdp_lqr_solve_batched            LQR.lqrSolver              (reference PDP/PDP.py:446-615)
//   pdp_cp_aux_integrate_batched     ControlPlanning.integrateAuxSys (PDP/PDP.py:813-838)
//   pdp_sysid_aux_integrate_batched  SysID.integrateAuxSys      (PDP/PDP.py:1241-1259)
// One wavefront (= one 64-thread workgroup) per trajectory; matrices are read from HBM straight into the
// register tile layout, every product is a chain of v_mfma_f64_16x16x4_f64 (pdp_tile.h / pdp_riccati.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../include/pdp_hip.h"
#include "pdp_lqr_kernels.h"
#include "pdp_lqr_stream_kernels.h"
#include <cstdlib>

using namespace pdp;

namespace {

constexpr int MAX_NT = 4;

inline int launched() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fprintf(stderr, "[pdp_hip] kernel launch failed: %s\n", hipGetErrorString(e));
    return PDP_E_LAUNCH;
}
#define PDP_CLEAR() (void)hipGetLastError()   // parameter tiles: p <= (16 - m) + 16 * (MAX_NT - 1)

// U_t = Ux X + Ue ; X+ = F X + G U.   Generic n <= 16, m <= 16.  A wavefront carries NT tiles of 16 parameter columns (grid.y covers
// the rest), so F, G and Ux are read once per NT tiles; the operands of step t+1 stream in while step t computes (see RunPtr).
template <int NT>
__global__ void __launch_bounds__(64) cp_aux_kernel(int B, int T, int n, int m, int p, const double* __restrict__ F, const double* __restrict__ G,
                                                     const double* __restrict__ Ux, const double* __restrict__ Ue, const double* __restrict__ X0,
                                                     double* __restrict__ Xo, double* __restrict__ Uo) {
    const int b = blockIdx.x, cbase = blockIdx.y * 16 * NT, lane = threadIdx.x;
    const d4 z = zero4();
    TileMap mX[NT], mU[NT];
    d4 X[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int w = max(0, min(16, p - cbase - 16 * j));          // 0: this tile is past the last column (maps all absent)
        mX[j] = make_dense_map<false>(n, w, p, 0, 0, lane);
        mU[j] = make_dense_map<false>(m, w, p, 0, 0, lane);
        X[j] = (X0 && w > 0) ? load_map(X0 + (int64_t)b * n * p + cbase + 16 * j, mX[j]) : z;
        store_map(Xo + (int64_t)b * (T + 1) * n * p + cbase + 16 * j, mX[j], X[j]);
    }
    const pdp_mat none = {nullptr, 0, 0}, aF = {F, (int64_t)T * n * n, n * n}, aG = {G, (int64_t)T * n * m, n * m},
                  aUx = {Ux, (int64_t)T * m * n, m * n};
    RunPtr qF = make_run(aF, make_dense_map<true>(n, n, n, 0, 0, lane), none, mX[0], b, 0),
           qG = make_run(aG, make_dense_map<true>(n, m, m, 0, 0, lane), none, mX[0], b, 0),       // m x n
           qUx = make_run(aUx, make_dense_map<true>(m, n, n, 0, 0, lane), none, mX[0], b, 0),    // n x m
           qUe[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const pdp_mat aUe = {Ue + cbase + 16 * j, (int64_t)T * m * p, m * p};
        qUe[j] = make_run(aUe, mU[j], none, mX[0], b, 0);
    }
    struct Tiles { d4 FT, GT, UxT, Ue[NT]; };
    auto request = [&](Tiles& q) {
        q.FT = load_run(qF, 1); q.GT = load_run(qG, 1); q.UxT = load_run(qUx, 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) q.Ue[j] = load_run(qUe[j], 1);
    };
    auto step = [&](int t, const Tiles& c, Tiles& nx) {
        if (t + 1 < T) request(nx);
        const int64_t bt = (int64_t)b * T + t;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            d4 U = mma_tn(c.UxT, X[j], c.Ue[j]);        // Ux X + Ue
            d4 Xn = mma_tn(c.FT, X[j], z);
            X[j] = mma_tn(c.GT, U, Xn);
            store_map(Uo + bt * m * p + cbase + 16 * j, mU[j], U);
            store_map(Xo + ((int64_t)b * (T + 1) + t + 1) * n * p + cbase + 16 * j, mX[j], X[j]);
        }
    };
    Tiles ta, tb;
    request(ta);
    int t = 0;
    for (; t + 1 < T; t += 2) { step(t, ta, tb); step(t + 1, tb, ta); }
    if (t < T) step(t, ta, tb);
}

__global__ void __launch_bounds__(64) sysid_aux_kernel(int B, int T, int n, int p, const double* __restrict__ F, const double* __restrict__ E,
                                                        const double* __restrict__ X0, double* __restrict__ Xo) {
    const int b = blockIdx.x, c0 = blockIdx.y * 16, lane = threadIdx.x;
    const int w = min(16, p - c0);
    const d4 z = zero4();
    const TileMap mX = make_dense_map<false>(n, w, p, 0, 0, lane);
    d4 X = X0 ? load_map(X0 + (int64_t)b * n * p + c0, mX) : z;
    store_map(Xo + (int64_t)b * (T + 1) * n * p + c0, mX, X);
    const pdp_mat none = {nullptr, 0, 0}, aF = {F, (int64_t)T * n * n, n * n}, aE = {E + c0, (int64_t)T * n * p, n * p};
    RunPtr qF = make_run(aF, make_dense_map<true>(n, n, n, 0, 0, lane), none, mX, b, 0), qE = make_run(aE, mX, none, mX, b, 0);
    struct Tiles { d4 FT, Et; };
    auto request = [&](Tiles& q) { q.FT = load_run(qF, 1); q.Et = load_run(qE, 1); };
    auto step = [&](int t, const Tiles& c, Tiles& nx) {
        if (t + 1 < T) request(nx);
        X = mma_tn(c.FT, X, c.Et);
        store_map(Xo + ((int64_t)b * (T + 1) + t + 1) * n * p + c0, mX, X);
    };
    Tiles ta, tb;
    request(ta);
    int t = 0;
    for (; t + 1 < T; t += 2) { step(t, ta, tb); step(t + 1, tb, ta); }
    if (t < T) step(t, ta, tb);
}

// grad[b][j] = sum_t dcx.X + dcu.U + dhx.X_T : one thread per (b, j), coalesced over j
__global__ void cp_grad_contract_kernel(int B, int T, int n, int m, int p, const double* __restrict__ dcx, const double* __restrict__ dcu,
                                        const double* __restrict__ dhx, const double* __restrict__ X, const double* __restrict__ U,
                                        double* __restrict__ grad) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (j >= p) return;
    double acc = 0.0;
    for (int t = 0; t < T; ++t) {
        const double* cx = dcx + ((int64_t)b * T + t) * n;
        const double* cu = dcu + ((int64_t)b * T + t) * m;
        const double* Xt = X + ((int64_t)b * (T + 1) + t) * n * p;
        const double* Ut = U + ((int64_t)b * T + t) * m * p;
        for (int i = 0; i < n; ++i) acc += cx[i] * Xt[i * p + j];
        for (int i = 0; i < m; ++i) acc += cu[i] * Ut[i * p + j];
    }
    const double* XT = X + ((int64_t)b * (T + 1) + T) * n * p;
    for (int i = 0; i < n; ++i) acc += dhx[(int64_t)b * n + i] * XT[i * p + j];
    grad[(int64_t)b * p + j] = acc;
}

// The parameter update of a data-parallel gradient-descent loop as ONE launch (pdp_gd_update_batched): column sums of [loss | grad] over the batch in a fixed order
// (thread (r, j) sums the rows b = r, r + R, .. of column j; the R partial sums are added in ascending r), theta <- theta - lr * mean gradient, the traces at the
// device-side iteration counter, and the health counters.  One workgroup: the iteration counter is read by every thread before thread 0 advances it.
__global__ void __launch_bounds__(1024) gd_update_kernel(int B, int p, const double* __restrict__ loss, const double* __restrict__ grad, int gs, const int32_t* __restrict__ status,
                                                         const int32_t* __restrict__ converged, const int32_t* __restrict__ iterations, double lr, double* __restrict__ theta,
                                                         double* __restrict__ dtheta, double* __restrict__ loss_trace, double* __restrict__ par_trace, long long trace_len,
                                                         long long* __restrict__ counters) {
    __shared__ double part[1024];
    __shared__ unsigned long long cnt[3];
    const int tid = threadIdx.x;
    int W = 1;
    while (W < p + 1) W <<= 1;
    const int R = 1024 / W, j = tid % W, r = tid / W;
    if (tid < 3) cnt[tid] = 0;
    double acc = 0.0;
    if (j < p) { for (int b = r; b < B; b += R) acc += grad[(int64_t)b * gs + j]; }
    else if (j == p) { for (int b = r; b < B; b += R) acc += loss[b]; }
    part[tid] = acc;
    unsigned long long unc = 0, trb = 0, nwt = 0;
    for (int b = tid; b < B; b += 1024) {
        if (converged) unc += converged[b] == 0;
        if (status) trb += status[b] != 0;
        if (iterations) nwt += (unsigned long long)iterations[b];
    }
    __syncthreads();
    if (unc) atomicAdd(&cnt[0], unc);
    if (trb) atomicAdd(&cnt[1], trb);
    if (nwt) atomicAdd(&cnt[2], nwt);
    const long long k = counters[0];
    if (tid <= p) {
        double sum = 0.0;
        for (int q = 0; q < R; ++q) sum += part[q * W + tid];
        const double mean = sum / (double)B;
        if (tid < p) {
            const double d = -lr * mean, th = theta[tid] + d;
            dtheta[tid] = d;
            theta[tid] = th;
            if (par_trace && k < trace_len) par_trace[k * p + tid] = th;
        } else if (loss_trace && k < trace_len) loss_trace[k] = mean;
    }
    __syncthreads();
    if (tid == 0) { counters[0] = k + 1; counters[1] += (long long)cnt[0]; counters[2] += (long long)cnt[1]; counters[3] += (long long)cnt[2]; }
}

// the same update for parameter vectors beyond one workgroup's width (p + 1 > 1024: a [64, 64] policy has 5316): a thread owns the columns tid, tid + 1024, ... and sums
// each over the batch in ascending order
__global__ void __launch_bounds__(1024) gd_update_wide_kernel(int B, int p, const double* __restrict__ loss, const double* __restrict__ grad, int gs, const int32_t* __restrict__ status,
                                                              const int32_t* __restrict__ converged, const int32_t* __restrict__ iterations, double lr, double* __restrict__ theta,
                                                              double* __restrict__ dtheta, double* __restrict__ loss_trace, double* __restrict__ par_trace, long long trace_len,
                                                              long long* __restrict__ counters) {
    __shared__ unsigned long long cnt[3];
    const int tid = threadIdx.x;
    if (tid < 3) cnt[tid] = 0;
    unsigned long long unc = 0, trb = 0, nwt = 0;
    for (int b = tid; b < B; b += 1024) {
        if (converged) unc += converged[b] == 0;
        if (status) trb += status[b] != 0;
        if (iterations) nwt += (unsigned long long)iterations[b];
    }
    __syncthreads();
    if (unc) atomicAdd(&cnt[0], unc);
    if (trb) atomicAdd(&cnt[1], trb);
    if (nwt) atomicAdd(&cnt[2], nwt);
    const long long k = counters[0];
    for (int j = tid; j <= p; j += 1024) {
        double sum = 0.0;
        if (j < p) { for (int b = 0; b < B; ++b) sum += grad[(int64_t)b * gs + j]; }
        else { for (int b = 0; b < B; ++b) sum += loss[b]; }
        const double mean = sum / (double)B;
        if (j < p) {
            const double d = -lr * mean, th = theta[j] + d;
            dtheta[j] = d;
            theta[j] = th;
            if (par_trace && k < trace_len) par_trace[k * p + j] = th;
        } else if (loss_trace && k < trace_len) loss_trace[k] = mean;
    }
    __syncthreads();
    if (tid == 0) { counters[0] = k + 1; counters[1] += (long long)cnt[0]; counters[2] += (long long)cnt[1]; counters[3] += (long long)cnt[2]; }
}

// ---- integrateAuxSys beyond one tile per matrix (ControlPlanning PDP.py:813-838, SysID PDP.py:1241-1259; n > 16): one LANE per (trajectory, parameter column), plain
// loops over global memory - U_t[:, j] = Ux_t X_t[:, j] + Ue_t[:, j], X_{t+1}[:, j] = F_t X_t[:, j] + G_t U_t[:, j] (+ E_t[:, j]).  The column of X_t is read back from
// the output array the same lane wrote.  Correctness first: these are the materialised drop-ins, the fused step kernels do not come through here.
__global__ void __launch_bounds__(64) cp_aux_generic_kernel(int B, int T, int n, int m, int p, const double* __restrict__ F, const double* __restrict__ G,
                                                            const double* __restrict__ Ux, const double* __restrict__ Ue, const double* __restrict__ X0,
                                                            double* __restrict__ X, double* __restrict__ U) {
    const int64_t id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (id >= (int64_t)B * p) return;
    const int b = (int)(id / p), j = (int)(id - (int64_t)b * p);
    double* Xb = X + (int64_t)b * (T + 1) * n * p;
    double* Ub = U + (int64_t)b * T * m * p;
    for (int i = 0; i < n; ++i) Xb[i * p + j] = X0 ? X0[((int64_t)b * n + i) * p + j] : 0.0;
    for (int t = 0; t < T; ++t) {
        const double *Ft = F + ((int64_t)b * T + t) * n * n, *Gt = G + ((int64_t)b * T + t) * n * m, *Uxt = Ux + ((int64_t)b * T + t) * m * n,
                     *Uet = Ue + ((int64_t)b * T + t) * m * p;
        const double* xc = Xb + (int64_t)t * n * p;
        double* xn = Xb + (int64_t)(t + 1) * n * p;
        double* ut = Ub + (int64_t)t * m * p;
        for (int r = 0; r < m; ++r) { double s = Uet[r * p + j]; for (int k = 0; k < n; ++k) s += Uxt[r * n + k] * xc[k * p + j]; ut[r * p + j] = s; }
        __threadfence_block();
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += Ft[i * n + k] * xc[k * p + j];
            for (int r = 0; r < m; ++r) s += Gt[i * m + r] * ut[r * p + j];
            xn[i * p + j] = s;
        }
        __threadfence_block();
    }
}
__global__ void __launch_bounds__(64) sysid_aux_generic_kernel(int B, int T, int n, int p, const double* __restrict__ F, const double* __restrict__ E,
                                                               const double* __restrict__ X0, double* __restrict__ X) {
    const int64_t id = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (id >= (int64_t)B * p) return;
    const int b = (int)(id / p), j = (int)(id - (int64_t)b * p);
    double* Xb = X + (int64_t)b * (T + 1) * n * p;
    for (int i = 0; i < n; ++i) Xb[i * p + j] = X0 ? X0[((int64_t)b * n + i) * p + j] : 0.0;
    for (int t = 0; t < T; ++t) {
        const double *Ft = F + ((int64_t)b * T + t) * n * n, *Et = E + ((int64_t)b * T + t) * n * p;
        const double* xc = Xb + (int64_t)t * n * p;
        double* xn = Xb + (int64_t)(t + 1) * n * p;
        for (int i = 0; i < n; ++i) { double s = Et[i * p + j]; for (int k = 0; k < n; ++k) s += Ft[i * n + k] * xc[k * p + j]; xn[i * p + j] = s; }
        __threadfence_block();
    }
}

template <int M>
int launch_lqr(const pdp_lqr_problem& pr, int nt, double* X, double* U, double* Lam, int32_t* status, double* wg, double* wpw, hipStream_t s) {
    dim3 grid(pr.B), block(64);
    PDP_CLEAR();
    switch (nt) {
        case 1: hipLaunchKernelGGL((lqr_solve_kernel<M, 1>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
        case 2: hipLaunchKernelGGL((lqr_solve_kernel<M, 2>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
        case 3: hipLaunchKernelGGL((lqr_solve_kernel<M, 3>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
        default: hipLaunchKernelGGL((lqr_solve_kernel<M, 4>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
    }
    return launched();
}

}  // namespace

extern "C" {

const char* pdp_hip_version(void) { return "pdp_hip 0.1 gfx950"; }

#ifdef PDP_LQS_TIMING      // probe builds only (probes/lqr_stream_timing.py)
int pdp_lqs_read_stamps(long long* out, void* stream) { hipLaunchKernelGGL(lqs_read_stamps, dim3(1), dim3(64), 0, (hipStream_t)stream, out); return launched(); }
#endif

// gains (+ P, W for the costate output) per stage; systems beyond one tile per matrix whose working set does not fit the LDS add their scratch per trajectory
static int64_t lqr_stage_doubles(int n, int m, int p, int want_costate) { return (int64_t)n * m + (int64_t)m * p + (want_costate ? (int64_t)n * n + (int64_t)n * p : 0); }
int64_t pdp_lqr_workspace_bytes(int B, int T, int n, int m, int p, int want_costate) {
    int64_t d = (int64_t)B * T * lqr_stage_doubles(n, m, p, want_costate);
    const int pl = p < GEN_PMAX ? p : GEN_PMAX;
    if ((n > 16 || m > 4) && !lqr_generic_in_lds(n, m, pl)) d += (int64_t)B * (int64_t)lqr_generic_lds_doubles(n, m, pl);
    return d * (int64_t)sizeof(double);
}

int pdp_lqr_solve_batched(const pdp_lqr_problem* prob, double* X, double* U, double* Lam, int32_t* status, void* workspace,
                          int64_t workspace_bytes, void* stream) {
    if (!prob || !X || !U || !workspace) return PDP_E_ARG;
    const pdp_lqr_problem& pr = *prob;
    if (pr.B <= 0 || pr.T <= 0 || pr.n <= 0 || pr.m <= 0 || pr.p <= 0) return PDP_E_ARG;
    if (!pr.F.ptr || !pr.G.ptr || !pr.Hxx.ptr || !pr.Huu.ptr || !pr.hxx.ptr) return PDP_E_ARG;
    if (pr.n > 16 || pr.m > 4) {                   // beyond one tile per matrix: the size-generic kernel (any n, m; p <= 32 per launch - the callers cut more into column blocks)
        if (pr.p > GEN_PMAX) return PDP_E_SIZE;
        if (workspace_bytes < pdp_lqr_workspace_bytes(pr.B, pr.T, pr.n, pr.m, pr.p, Lam != nullptr)) return PDP_E_ARG;
        double* wg = (double*)workspace;
        double* wpw = Lam ? wg + (int64_t)pr.B * pr.T * ((int64_t)pr.n * pr.m + (int64_t)pr.m * pr.p) : nullptr;
        double* wscr = wg + (int64_t)pr.B * pr.T * lqr_stage_doubles(pr.n, pr.m, pr.p, Lam != nullptr);
        PDP_CLEAR();
        if (lqr_generic_in_lds(pr.n, pr.m, pr.p)) {
            const size_t lds = sizeof(double) * lqr_generic_lds_doubles(pr.n, pr.m, pr.p);
            (void)hipFuncSetAttribute((const void*)lqr_solve_generic_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(lqr_solve_generic_kernel<false>, dim3(pr.B), dim3(64), lds, (hipStream_t)stream, pr, X, U, Lam, status, wg, wpw, (double*)nullptr);
        } else {
            hipLaunchKernelGGL(lqr_solve_generic_kernel<true>, dim3(pr.B), dim3(64), 0, (hipStream_t)stream, pr, X, U, Lam, status, wg, wpw, wscr);
        }
        return launched();
    }
    for (const pdp_mat* mt : {&pr.F, &pr.G, &pr.E, &pr.Hxx, &pr.Hxu, &pr.Hxe, &pr.Huu, &pr.Hue})       // per-lane byte strides are 32-bit
        if (mt->tstride < 0 || mt->tstride > (INT32_MAX >> 3)) return PDP_E_SIZE;
    const int p0 = pr.p < 16 - pr.m ? pr.p : 16 - pr.m;
    const int nt = 1 + (pr.p - p0 + 15) / 16;
    if (nt > MAX_NT) return PDP_E_SIZE;
    if (workspace_bytes < pdp_lqr_workspace_bytes(pr.B, pr.T, pr.n, pr.m, pr.p, Lam != nullptr)) return PDP_E_ARG;
    double* wg = (double*)workspace;
    double* wpw = Lam ? wg + (int64_t)pr.B * pr.T * (pr.n * pr.m + pr.m * pr.p) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (pr.n <= 4 && pr.m + pr.p <= 16) {          // small systems: four trajectories per wavefront, block-diagonal in the tile
        const dim3 grid((pr.B + 3) / 4), block(64);
        PDP_CLEAR();
        switch (pr.m) {
            case 1: hipLaunchKernelGGL((lqr_solve_small_kernel<1>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
            case 2: hipLaunchKernelGGL((lqr_solve_small_kernel<2>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
            case 3: hipLaunchKernelGGL((lqr_solve_small_kernel<3>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
            default: hipLaunchKernelGGL((lqr_solve_small_kernel<4>), grid, block, 0, s, pr, X, U, Lam, status, wg, wpw); break;
        }
        return launched();
    }
    // dense matrices from HBM, one parameter tile: the runner / streamer kernel (pdp_lqr_stream_kernels.h); PDP_LQR_VARIANT=1 keeps the one-wave kernel
    static const int variant = [] { const char* e = std::getenv("PDP_LQR_VARIANT"); return e ? std::atoi(e) : 2; }();
    if (variant == 2 && nt == 1 && lqs_ok(pr.n, pr.m, pr.p, Lam != nullptr)) {
        const dim3 grid((pr.B + 3) / 4), block(512);
        PDP_CLEAR();
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipLaunchKernelGGL(kern, grid, block, 160 * 1024, s, pr, X, U, Lam, status, wg, wpw);
        };
        const bool nl12 = lqs_lines(pr.n, pr.m, pr.p, Lam != nullptr) <= 12;      // lines per ring slot: 12 (C3 sizes and smaller) or 16
        switch (pr.m) {
            case 1: if (nl12) go(lqr_solve_stream_kernel<1, 12>); else go(lqr_solve_stream_kernel<1, 16>); break;
            case 2: if (nl12) go(lqr_solve_stream_kernel<2, 12>); else go(lqr_solve_stream_kernel<2, 16>); break;
            case 3: if (nl12) go(lqr_solve_stream_kernel<3, 12>); else go(lqr_solve_stream_kernel<3, 16>); break;
            default: if (nl12) go(lqr_solve_stream_kernel<4, 12>); else go(lqr_solve_stream_kernel<4, 16>); break;
        }
        return launched();
    }
    switch (pr.m) {
        case 1: return launch_lqr<1>(pr, nt, X, U, Lam, status, wg, wpw, s);
        case 2: return launch_lqr<2>(pr, nt, X, U, Lam, status, wg, wpw, s);
        case 3: return launch_lqr<3>(pr, nt, X, U, Lam, status, wg, wpw, s);
        default: return launch_lqr<4>(pr, nt, X, U, Lam, status, wg, wpw, s);
    }
}

int pdp_cp_aux_integrate_batched(int B, int T, int n, int m, int p, const double* F, const double* G, const double* Ux, const double* Ue,
                                 const double* X0, double* X, double* U, void* stream) {
    if (B <= 0 || T <= 0 || n <= 0 || m <= 0 || p <= 0 || !F || !G || !Ux || !Ue || !X || !U) return PDP_E_ARG;
    PDP_CLEAR();
    if (n > 16 || m > 16) {         // beyond one tile per matrix: lane per (trajectory, column), any size
        hipLaunchKernelGGL(cp_aux_generic_kernel, dim3((unsigned)(((int64_t)B * p + 63) / 64)), dim3(64), 0, (hipStream_t)stream, B, T, n, m, p, F, G, Ux, Ue, X0, X, U);
        return launched();
    }
    const int ntile = (p + 15) / 16;
    if (ntile == 1) hipLaunchKernelGGL(cp_aux_kernel<1>, dim3(B, 1), dim3(64), 0, (hipStream_t)stream, B, T, n, m, p, F, G, Ux, Ue, X0, X, U);
    else hipLaunchKernelGGL(cp_aux_kernel<2>, dim3(B, (ntile + 1) / 2), dim3(64), 0, (hipStream_t)stream, B, T, n, m, p, F, G, Ux, Ue, X0, X, U);
    return launched();
}

int pdp_cp_grad_contract_batched(int B, int T, int n, int m, int p, const double* dcx, const double* dcu, const double* dhx, const double* X,
                                 const double* U, double* grad, void* stream) {
    if (B <= 0 || T <= 0 || n <= 0 || m <= 0 || p <= 0 || !dcx || !dcu || !dhx || !X || !U || !grad) return PDP_E_ARG;
    PDP_CLEAR();
    hipLaunchKernelGGL(cp_grad_contract_kernel, dim3((p + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, B, T, n, m, p, dcx, dcu, dhx, X, U, grad);
    return launched();
}

int pdp_gd_update_batched(int B, int p, const double* loss, const double* grad, int grad_bstride, const int32_t* status, const int32_t* converged, const int32_t* iterations,
                          double lr, double* theta, double* dtheta, double* loss_trace, double* parameter_trace, int64_t trace_len, int64_t* counters, void* stream) {
    if (B <= 0 || p <= 0 || !loss || !grad || grad_bstride < p || !theta || !dtheta || !counters) return PDP_E_ARG;
    PDP_CLEAR();
    if (p + 1 > 1024) {
        hipLaunchKernelGGL(gd_update_wide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, p, loss, grad, grad_bstride, status, converged, iterations, lr, theta, dtheta,
                           loss_trace, parameter_trace, (long long)trace_len, (long long*)counters);
        return launched();
    }
    hipLaunchKernelGGL(gd_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, p, loss, grad, grad_bstride, status, converged, iterations, lr, theta, dtheta, loss_trace,
                       parameter_trace, (long long)trace_len, (long long*)counters);
    return launched();
}

int pdp_sysid_aux_integrate_batched(int B, int T, int n, int p, const double* F, const double* E, const double* X0, double* X, void* stream) {
    if (B <= 0 || T <= 0 || n <= 0 || p <= 0 || !F || !E || !X) return PDP_E_ARG;
    PDP_CLEAR();
    if (n > 16) {
        hipLaunchKernelGGL(sysid_aux_generic_kernel, dim3((unsigned)(((int64_t)B * p + 63) / 64)), dim3(64), 0, (hipStream_t)stream, B, T, n, p, F, E, X0, X);
        return launched();
    }
    hipLaunchKernelGGL(sysid_aux_kernel, dim3(B, (p + 15) / 16), dim3(64), 0, (hipStream_t)stream, B, T, n, p, F, E, X0, X);
    return launched();
}

}  // extern "C"
