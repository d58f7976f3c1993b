// pdp_model.hip - C-ABI entry points of one generated model library (libpdp_model_<name>.so), section B of
// include/pdp_hip.h.  Compiled once per model with -DPDP_MODEL_HEADER="generated/<name>.h" (codegen.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../include/pdp_hip.h"
#ifndef PDP_MODEL_HEADER
#error "compile with -DPDP_MODEL_HEADER=\"generated/<model>.h\""
#endif
#define PDP_HD __host__ __device__ inline
#include PDP_MODEL_HEADER
#ifdef PDP_PHASE_TIMING_FINE      // timing builds with -DPDP_PHASE_TIMING_FINE: cycle stamps inside the Riccati step of the fused3 runner (probes/phase_timing3.py)
namespace pdp { extern __device__ long long g_rb_stamp[16]; }
#define PDP_RB_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) pdp::g_rb_stamp[i] = __builtin_readcyclecounter(); } while (0)
#endif
#include "pdp_model_kernels.h"
#include "pdp_lqr_kernels.h"
#include "pdp_ocsolve_kernels.h"
#include "pdp_ocsolve2_kernels.h"
#include "pdp_cp_mlp_kernels.h"
#include "pdp_fused3_kernels.h"
#include "pdp_cp_pair_kernels.h"
#include "pdp_cp_generic_kernels.h"
#include <cstdlib>

using namespace pdp;

#ifndef PDP_FUSED_DEFAULT_VARIANT
#define PDP_FUSED_DEFAULT_VARIANT 3
#endif

namespace {

// launch-error protocol: stale errors of other libraries in the process are cleared on entry (PDP_CLEAR), the error
// of our own launch is reported on stderr and mapped to PDP_E_LAUNCH
inline int launched() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    fprintf(stderr, "[pdp_hip] kernel launch failed: %s\n", hipGetErrorString(e));
    return PDP_E_LAUNCH;
}
#define PDP_CLEAR() (void)hipGetLastError()
inline hipStream_t S(void* s) { return (hipStream_t)s; }

[[maybe_unused]] inline int device_cu_count() {
    static int n = 0;
    if (n == 0) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
    return n;
}

template <class Mdl> constexpr bool fused_oc_ok() { return Mdl::KIND == PDP_KIND_OC && Mdl::NX <= 16 && Mdl::NU <= 4 && Mdl::NU + Mdl::NP <= 16; }

template <class Mdl>
int64_t oc_ws_bytes(int B, int T) {
    return (int64_t)B * T * fused_gain_doubles<Mdl>() * (int64_t)sizeof(double);
}

template <class Mdl>
int oc_rollout(int B, int T, const double* x0, const double* u, const double* th, int tb, double* x, double* cost, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        if (B <= 0 || T <= 0 || !x0 || !u || !th || !x) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_rollout_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, x0, u, th, tb, x, cost);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int oc_rollout_fb(int B, int T, const double* x0, const double* ubar, const double* xbar, const double* gains, const double* alpha, const double* th,
                  int tb, double* x, double* u, double* cost, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        if (B <= 0 || T <= 0 || !x0 || !ubar || !xbar || !gains || !alpha || !th || !x || !u || !cost) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_rollout_feedback_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, x0, ubar, xbar, gains, alpha, th, tb, x, u, cost);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int oc_ms_residuals(int B, int T, const double* x, const double* u, const double* lam, const double* th, int tb, double* c, double* rx, double* ru, double* cost,
                    void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        if (B <= 0 || T <= 0 || !x || !u || !lam || !th || !c || !rx || !ru || !cost) return PDP_E_ARG;
        PDP_CLEAR();
        const int64_t nthr = (int64_t)B * (T + 1);
        hipLaunchKernelGGL((oc_ms_residuals_kernel<Mdl>), dim3((unsigned)((nthr + 63) / 64)), dim3(64), 0, S(st), B, T, x, u, lam, th, tb, c, rx, ru, cost);
        return launched();
    } else return PDP_E_MODE;
}
template <class Mdl>
int oc_costate(int B, int T, const double* x, const double* u, const double* th, int tb, double* lam, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        if (B <= 0 || T <= 0 || !x || !u || !th || !lam) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_costate_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, x, u, th, tb, lam);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int oc_auxsys(int B, int T, const double* x, const double* u, const double* lam, const double* th, int tb, const pdp_oc_auxsys* o, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        if (B <= 0 || T <= 0 || !x || !u || !lam || !th || !o) return PDP_E_ARG;
        const int nchunk = (T + auxsys_chunk<Mdl>() - 1) / auxsys_chunk<Mdl>();
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_auxsys_kernel<Mdl>), dim3((unsigned)((int64_t)B * (nchunk + 1))), dim3(64), 0, S(st), B, T, x, u, lam, th, tb, *o);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int oc_predict(int B, int T, const double* dth, int dtb, const double* dxdp, const double* dudp, const double* ric, double* x, double* u, double* lam, void* st) {
    if constexpr (fused_oc_ok<Mdl>()) {
        if (B <= 0 || T <= 0 || !dth || !dxdp || !dudp || !x || !u || (ric && !lam)) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_predict_kernel<Mdl>), dim3((unsigned)((int64_t)B * ((T + 3) / 4))), dim3(64), 0, S(st), B, T, dth, dtb, dxdp, dudp, ric, x, u, lam);
        return launched();
    } else { return Mdl::KIND == PDP_KIND_OC ? PDP_E_SIZE : PDP_E_MODE; }
}

template <class Mdl>
int oc_predict_rec(int B, int T, const double* dth, int dtb, const float* rec, double* x, double* u, double* lam, void* st) {
    if constexpr (fused_oc_ok<Mdl>()) {
        if (B <= 0 || T <= 0 || !dth || !rec || !x || !u) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_predict_rec_kernel<Mdl>), dim3((unsigned)((int64_t)B * ((T + 3) / 4))), dim3(64), 0, S(st), B, T, dth, dtb, rec, x, u, lam);
        return launched();
    } else { return Mdl::KIND == PDP_KIND_OC ? PDP_E_SIZE : PDP_E_MODE; }
}
template <class Mdl>
int oc_pdp(int B, int T, int flags, const double* x0, const double* u, const double* th, int tb, const double* dx, const double* du, double* x,
           double* lam, double* loss, double* grad, double* dxdp, double* dudp, double* ric, float* prec, int32_t* status, void* ws, int64_t wsb, void* st) {
    if constexpr (fused_oc_ok<Mdl>()) {
        if (B <= 0 || T <= 0 || !u || !th || !dx || !du || !x || !lam || !loss || !grad || !ws) return PDP_E_ARG;
        if (!(flags & PDP_OC_GIVEN_TRAJ) && !x0) return PDP_E_ARG;
        if (wsb < oc_ws_bytes<Mdl>(B, T)) return PDP_E_ARG;
        const size_t lds = fused_lds_bytes<Mdl>(T);
        if (lds > 160 * 1024) return PDP_E_SIZE;
        // Kernel variants (environment PDP_FUSED_VARIANT overrides the default): 3 = runner / evaluator wave pair per trajectory, four
        // trajectories per 512-thread workgroup (pdp_fused3_kernels.h) - the default wherever it applies (n > 4, rollout staging within the
        // pool area); 1 = one wavefront per trajectory (systems with n <= 4, long horizons)
        static const int variant = [] { const char* e = std::getenv("PDP_FUSED_VARIANT"); return e ? std::atoi(e) : PDP_FUSED_DEFAULT_VARIANT; }();
        PDP_CLEAR();
        if constexpr (Mdl::NX > 4) {
            if (variant == 3 && fused3_ok<Mdl>(T)) {
                // trajectories per workgroup: 4 (runner and evaluator share a SIMD) once the batch fills the chip that way; a smaller batch spreads over
                // the CUs with the two waves of a trajectory on different SIMDs (PDP_FUSED_TPW overrides)
                static const int tpw_env = [] { const char* e = std::getenv("PDP_FUSED_TPW"); return e ? std::atoi(e) : 0; }();
                const int cus = device_cu_count();
                const int tpw = tpw_env ? tpw_env : (B <= cus ? 1 : (B <= 2 * cus ? 2 : 4));
                auto go = [&](auto kern, int TPW) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TPW * 40 * 1024);
                    hipLaunchKernelGGL(kern, dim3((B + TPW - 1) / TPW), dim3(128 * TPW), TPW * 40 * 1024, S(st), B, T, flags, x0, u, th, tb, dx, du, x, lam, loss, grad, dxdp,
                                       dudp, status, (double*)ws, ric, prec);
                    return launched();
                };
                if (ric || dxdp || dudp || prec) {          // sensitivity outputs (any of dxdp, dudp, the Riccati record): the instantiation that writes them with buffer stores
                    if (tpw == 1) return go(oc_pdp_fused3_kernel<Mdl, 1, true>, 1);
                    if (tpw == 2) return go(oc_pdp_fused3_kernel<Mdl, 2, true>, 2);
                    return go(oc_pdp_fused3_kernel<Mdl, 4, true>, 4);
                }
                if (tpw == 1) return go(oc_pdp_fused3_kernel<Mdl, 1>, 1);
                if (tpw == 2) return go(oc_pdp_fused3_kernel<Mdl, 2>, 2);
                return go(oc_pdp_fused3_kernel<Mdl, 4>, 4);
            }
        }
        auto go1 = [&](auto kern) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(B), dim3(64), lds, S(st), B, T, flags, x0, u, th, tb, dx, du, x, lam, loss, grad, dxdp, dudp, status, (double*)ws, ric, prec);
            return launched();
        };
        return (ric || prec) ? go1(oc_pdp_fused_kernel<Mdl, true>) : go1(oc_pdp_fused_kernel<Mdl, false>);
    } else { return Mdl::KIND == PDP_KIND_OC ? PDP_E_SIZE : PDP_E_MODE; }
}

// ---- batched Newton solve (pdp_oc_solve_batched): workspace carve-up and the iteration loop ---------------------------------
template <class Mdl>
struct OcSolveWs {
    double *lam_eff, *F, *G, *Hxx, *Hxu, *Huu, *hxx, *dHu, *hxe0, *dX, *dU, *lqr, *xt, *ut, *Jt;
    OcSolveState st;
    int64_t bytes;
    OcSolveWs(void* base, int B, int T, int K) {
        constexpr int n = Mdl::NX, m = Mdl::NU;
        char* p = (char*)base;
        int64_t off = 0;
        auto take = [&](int64_t count, int64_t elem) { void* r = p ? p + off : nullptr; off += (count * elem + 255) / 256 * 256; return r; };
        const int64_t BT = (int64_t)B * T;
        lam_eff = (double*)take(BT * n, 8); F = (double*)take(BT * n * n, 8); G = (double*)take(BT * n * m, 8);
        Hxx = (double*)take(BT * n * n, 8); Hxu = (double*)take(BT * n * m, 8); Huu = (double*)take(BT * m * m, 8);
        hxx = (double*)take((int64_t)B * n * n, 8); dHu = (double*)take(BT * m, 8); hxe0 = (double*)take((int64_t)B * n, 8);
        dX = (double*)take((int64_t)B * (T + 1) * n, 8); dU = (double*)take(BT * m, 8); lqr = (double*)take(BT * (n * m + m), 8);
        xt = (double*)take((int64_t)B * K * (T + 1) * n, 8); ut = (double*)take((int64_t)B * K * T * m, 8); Jt = (double*)take((int64_t)B * K, 8);
        st.J = (double*)take(B, 8); st.mu = (double*)take(B, 8); st.gnorm = (double*)take(B, 8);
        st.newton = (int32_t*)take(B, 4); st.converged = (int32_t*)take(B, 4); st.lqr_status = (int32_t*)take(B, 4); st.counters = (int32_t*)take(2, 4);
        bytes = off;
    }
};

template <class Mdl>
int oc_solve(int B, int T, const double* x0, const double* th, int tb, double* u, double* x, double* lam, double* cost, double* grad_norm,
             int32_t* converged, double* gains, const pdp_oc_solve_opts* op, int* iterations, void* ws, int64_t wsb, void* stv) {
    if constexpr (Mdl::KIND == PDP_KIND_OC && lqr_generic_in_lds(Mdl::NX, Mdl::NU, 1)) {      // beyond n = 16 / m = 4 the LQ step runs on the size-generic kernel (working set in LDS: n up to ~75)
        constexpr int n = Mdl::NX, m = Mdl::NU;
        if (B <= 0 || T <= 0 || !x0 || !th || !u || !x || !lam || !op || !ws) return PDP_E_ARG;
        const int K = op->ls_trials > 0 ? op->ls_trials : 10, every = op->check_every > 0 ? op->check_every : 4;
        OcSolveWs<Mdl> w(ws, B, T, K);
        if (wsb < w.bytes) return PDP_E_ARG;
        hipStream_t st = S(stv);
        const int nchunk = (T + auxsys_chunk<Mdl>() - 1) / auxsys_chunk<Mdl>();
        const dim3 gaux((unsigned)((int64_t)B * (nchunk + 1))), gB((B + 63) / 64), gBK((B * K + 63) / 64);
        PDP_CLEAR();
        (void)hipMemsetAsync(w.hxe0, 0, sizeof(double) * B * n, st);
        (void)hipMemsetAsync(w.st.mu, 0, sizeof(double) * B, st);
        (void)hipMemsetAsync(w.st.newton, 0, sizeof(int32_t) * B, st);
        (void)hipMemsetAsync(w.st.counters, 0, sizeof(int32_t) * 2, st);
        hipLaunchKernelGGL((oc_rollout_kernel<Mdl>), gB, dim3(64), 0, st, B, T, x0, u, th, tb, x, w.st.J);
        // LQ sub-problem for (dx, du): the LQR.lqrSolver kernel with p = 1, Hue := H_u, E = Hxe = 0 (PDP.py:557-608)
        pdp_lqr_problem pr{};
        pr.B = B; pr.T = T; pr.n = n; pr.m = m; pr.p = 1;
        pr.F = {w.F, (int64_t)T * n * n, n * n}; pr.G = {w.G, (int64_t)T * n * m, n * m}; pr.Hxx = {w.Hxx, (int64_t)T * n * n, n * n};
        pr.Hxu = {w.Hxu, (int64_t)T * n * m, n * m}; pr.Huu = {w.Huu, (int64_t)T * m * m, m * m}; pr.Hue = {w.dHu, (int64_t)T * m, m};
        pr.hxx = {w.hxx, n * n, 0}; pr.hxe = {w.hxe0, n, 0};
        auto lq = [&]() {
            if constexpr (n <= 4 && m <= 4)       // small systems: four trajectories per wavefront (pdp_riccati_small.h; its tile rows hold m <= 4 controls)
                hipLaunchKernelGGL((lqr_solve_small_kernel<m>), dim3((B + 3) / 4), dim3(64), 0, st, pr, w.dX, w.dU, (double*)nullptr, w.st.lqr_status, w.lqr, (double*)nullptr);
            else if constexpr (n <= 16 && m <= 4)
                hipLaunchKernelGGL((lqr_solve_kernel<m, 1>), dim3(B), dim3(64), 0, st, pr, w.dX, w.dU, (double*)nullptr, w.st.lqr_status, w.lqr, (double*)nullptr);
            else {
                const size_t lds = sizeof(double) * lqr_generic_lds_doubles(n, m, 1);
                (void)hipFuncSetAttribute((const void*)lqr_solve_generic_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(lqr_solve_generic_kernel<false>, dim3(B), dim3(64), lds, st, pr, w.dX, w.dU, (double*)nullptr, w.st.lqr_status, w.lqr, (double*)nullptr,
                                   (double*)nullptr);
            }
        };
        pdp_oc_auxsys only_hu{}, hess{};
        only_hu.dHu = w.dHu;
        hess.dynF = w.F; hess.dynG = w.G; hess.Hxx = w.Hxx; hess.Hxu = w.Hxu; hess.Huu = w.Huu; hess.hxx = w.hxx; hess.Huu_damp = w.st.mu;
        int it = 0, last_nconv = 0, last_gain = 0, nconv = 0;
        for (it = 0; it < op->max_iter; ++it) {
            hipLaunchKernelGGL((oc_costate_kernel<Mdl>), gB, dim3(64), 0, st, B, T, x, u, th, tb, lam);
            hipLaunchKernelGGL((oc_auxsys_kernel<Mdl>), gaux, dim3(64), 0, st, B, T, x, u, lam, th, tb, only_hu);
            hipLaunchKernelGGL((oc_newton_prepare_kernel<Mdl>), dim3(B), dim3(64), 0, st, B, T, it, op->tol, op->newton_switch, u, w.dHu, lam, w.lam_eff, w.st);
            if (it % every == 0 || op->print_level > 0) {                 // poll the number of converged samples (synchronises the stream)
                int32_t c = 0;
                if (hipMemcpyAsync(&c, &w.st.counters[it & 1], sizeof(c), hipMemcpyDeviceToHost, st) != hipSuccess) return PDP_E_LAUNCH;
                if (hipStreamSynchronize(st) != hipSuccess) return launched() ? PDP_E_LAUNCH : PDP_E_LAUNCH;
                nconv = c;
                if (op->print_level > 0) fprintf(stderr, "  pdp_oc_solve iter %3d  converged %d/%d\n", it, nconv, B);
                if (nconv == B) break;
                // stragglers: most of the batch done and nothing new for a while -> stop, the caller re-solves the rest from a neighbour
                if (nconv > last_nconv) { last_nconv = nconv; last_gain = it; }
                else if (op->straggler_patience > 0 && nconv >= 0.9 * B && it - last_gain >= op->straggler_patience) break;
            }
            hipLaunchKernelGGL((oc_auxsys_kernel<Mdl>), gaux, dim3(64), 0, st, B, T, x, u, w.lam_eff, th, tb, hess);
            lq();
            hipLaunchKernelGGL((oc_linesearch_kernel<Mdl>), gBK, dim3(64), 0, st, B, T, K, x0, u, x, w.lqr, th, tb, w.xt, w.ut, w.Jt);
            hipLaunchKernelGGL((oc_ls_select_kernel<Mdl>), dim3(B), dim3(64), 0, st, B, T, K, w.dHu, w.dU, w.xt, w.ut, w.Jt, x, u, w.st);
        }
        if (it == op->max_iter) hipLaunchKernelGGL((oc_costate_kernel<Mdl>), gB, dim3(64), 0, st, B, T, x, u, th, tb, lam);
        if (gains) {       // time-varying LQR feedback around the final trajectory (full Hamiltonian Hessians, no damping)
            hess.Huu_damp = nullptr;
            hipLaunchKernelGGL((oc_auxsys_kernel<Mdl>), gaux, dim3(64), 0, st, B, T, x, u, lam, th, tb, only_hu);
            hipLaunchKernelGGL((oc_auxsys_kernel<Mdl>), gaux, dim3(64), 0, st, B, T, x, u, lam, th, tb, hess);
            lq();
            (void)hipMemcpyAsync(gains, w.lqr, sizeof(double) * (int64_t)B * T * (n * m + m), hipMemcpyDeviceToDevice, st);
        }
        if (cost) (void)hipMemcpyAsync(cost, w.st.J, sizeof(double) * B, hipMemcpyDeviceToDevice, st);
        if (grad_norm) (void)hipMemcpyAsync(grad_norm, w.st.gnorm, sizeof(double) * B, hipMemcpyDeviceToDevice, st);
        if (converged) (void)hipMemcpyAsync(converged, w.st.converged, sizeof(int32_t) * B, hipMemcpyDeviceToDevice, st);
        if (iterations) *iterations = it;
        return launched();
    } else { return Mdl::KIND == PDP_KIND_OC ? PDP_E_SIZE : PDP_E_MODE; }
}

// Multiple-shooting solver variants (environment PDP_MS_VARIANT overrides): 2 = runner / evaluator wave pair per trajectory
// (pdp_ocsolve2_kernels.h), the default wherever its LDS layout fits; 1 = one wavefront per trajectory (pdp_ocsolve_kernels.h).
inline int ms_variant() { static const int v = [] { const char* e = std::getenv("PDP_MS_VARIANT"); return e ? std::atoi(e) : 2; }(); return v; }
template <class Mdl>
int64_t oc_solve_ms_ws_bytes(int B, int T, int max_iter) {
    if constexpr (Mdl::KIND == PDP_KIND_OC) {
        const int64_t a = (int64_t)B * MsLayout<Mdl>::ws_doubles(T, max_iter < 0 ? 0 : max_iter) * (int64_t)sizeof(double);
        const int64_t c = ms2_ws_bytes<Mdl>(B, T, max_iter);
        return a > c ? a : c;                       // either variant may serve the call
    } else return 0;
}
template <class Mdl, int TPW, bool WD = false>
int oc_solve_ms2_launch(int B, int T, const pdp_oc_ms_opts* op, const double* x0, const double* th, int tb, double* x, double* u, double* lam, double* cost,
                        double* resid, int32_t* converged, int32_t* iterations, int32_t* status, double* gains, double* iter_log, void* ws, void* st) {
    constexpr int lds = TPW * Ms2Layout<Mdl>::SLICE * (int)sizeof(double);
    (void)hipFuncSetAttribute((const void*)oc_solve_ms2_kernel<Mdl, TPW, WD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    PDP_CLEAR();
    hipLaunchKernelGGL((oc_solve_ms2_kernel<Mdl, TPW, WD>), dim3((B + TPW - 1) / TPW), dim3(128 * TPW), lds, S(st), B, T, *op, x0, th, tb, x, u, lam, cost, resid,
                       converged, iterations, status, gains, op->log_rows > 0 ? iter_log : (double*)nullptr, (double*)ws);
    return launched();
}
template <class Mdl>
int oc_solve_ms(int B, int T, const double* x0, const double* th, int tb, double* x, double* u, double* lam, double* cost, double* resid,
                int32_t* converged, int32_t* iterations, int32_t* status, double* gains, double* iter_log, const pdp_oc_ms_opts* op, void* ws, int64_t wsb,
                void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_OC && Mdl::NX <= 16 && Mdl::NU <= 4) {
        if (B <= 0 || T <= 0 || !x0 || !th || !x || !u || !lam || !op || !ws) return PDP_E_ARG;
        if (op->max_iter < 0 || wsb < oc_solve_ms_ws_bytes<Mdl>(B, T, op->max_iter)) return PDP_E_ARG;
        const bool needs_pair = (op->flags & PDP_MS_FROM_CONTROLS) != 0;       // (the one-wave kernel has no restoration pass to start from)
        if (needs_pair && !(op->flags & PDP_MS_WARM)) return PDP_E_ARG;
        const bool watchdog = (op->flags & PDP_MS_WITH_WATCHDOG) != 0;
        const bool predict = (op->flags & PDP_MS_PREDICT) != 0;
        if ((op->flags & PDP_MS_PREDICT_PRIMAL) && (!predict || !op->predict_record)) return PDP_E_ARG;
        if (predict && (!(op->flags & PDP_MS_WARM) || needs_pair || !op->dtheta || (!op->predict_record && (!op->dxdp || !op->dudp)))) return PDP_E_ARG;
        // PDP_MS_PREDICT is applied by the runner / evaluator kernel while it loads the point (dx parked in its LDS pool); where that kernel does not run, or the
        // horizon outgrows the pool, the prediction is a launch of its own in front of the solve (pdp_oc_predict_batched, in place on x, u, lam)
        pdp_oc_ms_opts op1 = *op;
        auto predict_first = [&]() -> int {
            if (!predict) return 0;
            op1.flags &= ~PDP_MS_PREDICT;
            if constexpr (Mdl::NU + Mdl::NP <= 16) {
                if (op->predict_record)
                    return oc_predict_rec<Mdl>(B, T, op->dtheta, op->dtheta_bstride, op->predict_record, x, u, (op->flags & PDP_MS_PREDICT_PRIMAL) ? nullptr : lam, st);
                return oc_predict<Mdl>(B, T, op->dtheta, op->dtheta_bstride, op->dxdp, op->dudp, op->riccati, x, u, op->riccati ? lam : nullptr, st);
            } else return PDP_E_SIZE;
        };
        if constexpr (ms2_ok<Mdl>()) {
            if (ms_variant() == 2 || needs_pair) {
                if (predict && !op->predict_record && !Ms2Layout<Mdl>::predict_fits(T)) { const int rc = predict_first(); if (rc != 0) return rc; }      // (the record is staged block by block: any horizon)
                op = &op1;
                // trajectories per workgroup: 4 (runner and evaluator of a trajectory share a SIMD) once the batch fills the chip that way; smaller
                // batches spread over the CUs with the two waves of a trajectory on different SIMDs
                const int cus = device_cu_count();
                // PDP_MS_WITH_WATCHDOG: the instantiations with the watchdog, one / two trajectories per workgroup whatever the batch (see the kernel's template line)
                if (watchdog) {
                    if (B <= cus) return oc_solve_ms2_launch<Mdl, 1, true>(B, T, op, x0, th, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, ws, st);
                    return oc_solve_ms2_launch<Mdl, 2, true>(B, T, op, x0, th, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, ws, st);
                }
                if (B <= cus) return oc_solve_ms2_launch<Mdl, 1>(B, T, op, x0, th, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, ws, st);
                if (B <= 2 * cus) return oc_solve_ms2_launch<Mdl, 2>(B, T, op, x0, th, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, ws, st);
                return oc_solve_ms2_launch<Mdl, 4>(B, T, op, x0, th, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, ws, st);
            }
        }
        if (needs_pair || watchdog) return PDP_E_SIZE;      // (only the runner / evaluator kernel restores, and only it has the watchdog)
        const size_t lds = ms_lds_bytes<Mdl>();
        if (lds > 160 * 1024) return PDP_E_SIZE;
        { const int rc = predict_first(); if (rc != 0) return rc; }
        op = &op1;
        (void)hipFuncSetAttribute((const void*)oc_solve_ms_kernel<Mdl>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        PDP_CLEAR();
        hipLaunchKernelGGL((oc_solve_ms_kernel<Mdl>), dim3(B), dim3(64), lds, S(st), B, T, *op, x0, th, tb, x, u, lam, cost, resid, converged, iterations,
                           status, gains, op->log_rows > 0 ? iter_log : (double*)nullptr, (double*)ws);
        return launched();
    } else { return Mdl::KIND == PDP_KIND_OC ? PDP_E_SIZE : PDP_E_MODE; }
}

template <class Mdl> bool cp_policy_args_ok(const pdp_policy* pol, int p);
template <class Mdl>
int cp_integrate(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* x, double* u, double* cost, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_CP) {
        if (B <= 0 || T <= 0 || !pol || !x0 || !th) return PDP_E_ARG;
        // (networks beyond the lane-local arrays of this lane-per-trajectory integrator: PDP_E_SIZE tells the caller to take pdp_cp_step_batched's size-generic kernel,
        // which rolls out as well - runtime.cp_integrate does)
        if (pol->kind == PDP_POLICY_MLP) { if (pol->n_layers > 8) return PDP_E_SIZE; for (int k = 0; k < pol->n_layers; ++k) if (pol->sizes[k] > MLP_MAX_WIDTH) return PDP_E_SIZE; }
        PDP_CLEAR();
        hipLaunchKernelGGL((cp_integrate_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, *pol, p, x0, th, tb, x, u, cost);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int cp_auxsys(int B, int T, const pdp_policy* pol, int p, const double* x, const double* u, const double* th, int tb, double* F, double* G,
              double* Ux, double* Ue, double* cx, double* cu, double* hx, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_CP) {
        if (B <= 0 || T <= 0 || !pol || !x || !u || !th) return PDP_E_ARG;
        bool wide = false;                       // a network beyond the lane-local arrays of cp_auxsys_kernel: its Jacobians come from the wave-per-(b, t) kernel
        if (pol->kind == PDP_POLICY_MLP) { wide = pol->n_layers > 8; for (int k = 0; k < pol->n_layers && k < GEN_MAXL; ++k) wide = wide || pol->sizes[k] > MLP_MAX_WIDTH; }
        const int64_t n = (int64_t)B * (T + 1);
        PDP_CLEAR();
        hipLaunchKernelGGL((cp_auxsys_kernel<Mdl>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, S(st), B, T, *pol, p, x, u, th, tb, F, G, wide ? nullptr : Ux,
                           wide ? nullptr : Ue, cx, cu, hx);
        if (wide && Ux && Ue) {
            if (!cp_policy_args_ok<Mdl>(pol, p)) return PDP_E_ARG;
            int sum_in = Mdl::NX, maxw = Mdl::NX;
            for (int k = 0; k < pol->n_layers; ++k) { if (k + 1 < pol->n_layers) sum_in += pol->sizes[k]; maxw = pol->sizes[k] > maxw ? pol->sizes[k] : maxw; }
            const size_t lds = sizeof(double) * ((size_t)sum_in + 2 * (size_t)maxw + 8);
            if (lds > 150 * 1024) return PDP_E_SIZE;           // (layer inputs of more than ~19 000 units in total)
            (void)hipFuncSetAttribute((const void*)cp_policy_jac_generic_kernel<Mdl::NX, Mdl::NU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((cp_policy_jac_generic_kernel<Mdl::NX, Mdl::NU>), dim3((unsigned)((int64_t)B * T)), dim3(64), lds, S(st), B, T, *pol, p, x, th, tb, Ux, Ue,
                               sum_in, maxw);
        }
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl, int NT>
int cp_step_launch(int B, int gy, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* loss, double* grad, double* x,
                   double* u, void* st) {
    const size_t lds = sizeof(double) * (1 + Mdl::PATH_NCONST + Mdl::CHUNK * (Mdl::PATH_NVAR | 1) + (size_t)(T + 1) * Mdl::NX + (size_t)T * Mdl::NU +
                                         (size_t)T * pol->n_pivots + Mdl::NX + 8 + 64 + (Mdl::NX > Mdl::NU ? Mdl::NX : Mdl::NU));
    if (lds > 150 * 1024) return PDP_E_SIZE;
    (void)hipFuncSetAttribute((const void*)cp_step_poly_kernel<Mdl, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    PDP_CLEAR();
    hipLaunchKernelGGL((cp_step_poly_kernel<Mdl, NT>), dim3(B, gy), dim3(64), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, 0, (const double*)nullptr,
                       (const double*)nullptr, (const double*)nullptr);
    return launched();
}
// Batches from which ControlPlanning.step with the Lagrange policy rolls out beforehand, one LANE per trajectory, and runs the sensitivity kernel on the given trajectories
// (cp_poly_rollout_lanes_kernel + cp_step_poly_kernel<.., GIVEN>): more than two trajectories per SIMD, like SysID.step.  PDP_CP_PREPASS=0 / 1 forces it off / on.
inline bool cp_prepass(int B) {
    static const int env = [] { const char* e = std::getenv("PDP_CP_PREPASS"); return e ? std::atoi(e) : -1; }();
    return env >= 0 ? env != 0 : B > 8 * device_cu_count();
}
// workspace of the pre-pass: x [B][T+1][NX] | u [B][T][NU] | h_x [B][NX]
template <class Mdl>
int64_t cp_prepass_ws_bytes(int B, int T) { return (int64_t)B * ((int64_t)(T + 1) * Mdl::NX + (int64_t)T * Mdl::NU + Mdl::NX) * (int64_t)sizeof(double); }
template <class Mdl, int NT>
int cp_step_given_launch(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* loss, double* grad, double* x, double* u,
                         double* ws, void* st) {
    constexpr int NX = Mdl::NX, NU = Mdl::NU, STRIDE = Mdl::PATH_NVAR | 1;
    const int np = pol->n_pivots;
    double* xw = x ? x : ws;                                                   // the API outputs double as the hand-over where the caller asked for them
    double* uw = u ? u : ws + (int64_t)B * (T + 1) * NX;
    double* hxw = ws + (int64_t)B * ((int64_t)(T + 1) * NX + (int64_t)T * NU);
    const size_t lds0 = sizeof(double) * ((size_t)T * np + (size_t)p * 64);
    if (lds0 > 150 * 1024) return PDP_E_SIZE;
    (void)hipFuncSetAttribute((const void*)cp_poly_rollout_lanes_kernel<Mdl>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0);
    PDP_CLEAR();
    hipLaunchKernelGGL((cp_poly_rollout_lanes_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), lds0, S(st), B, T, *pol, p, x0, th, tb, loss, xw, uw, hxw);
    if (const int rc = launched(); rc != 0) return rc;
    static const int wgs_env = [] { const char* e = std::getenv("PDP_CP_GIVEN_WGS"); return e ? std::atoi(e) : 0; }();
    const int wgs = wgs_env > 0 ? wgs_env : 12;
    const int fixed = 1 + Mdl::PATH_NCONST + T * np + NX + 8 + 64 + (NX > NU ? NX : NU);
    int rows = (160 * 1024 / 8 / wgs - 64 - fixed) / STRIDE;
    rows = rows > Mdl::CHUNK ? Mdl::CHUNK : (rows < 4 ? (Mdl::CHUNK < 4 ? Mdl::CHUNK : 4) : rows);
    const size_t lds = sizeof(double) * ((size_t)fixed + (size_t)rows * STRIDE);
    if (lds > 150 * 1024) return PDP_E_SIZE;
    (void)hipFuncSetAttribute((const void*)cp_step_poly_kernel<Mdl, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((cp_step_poly_kernel<Mdl, NT, true>), dim3(B, 1), dim3(64), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, rows, (const double*)xw,
                       (const double*)uw, (const double*)hxw);
    return launched();
}
// Lagrange-policy kernel variants (environment PDP_CP_POLY_VARIANT overrides): 2 = rollout wave + sensitivity wave per trajectory (pdp_cp_pair_kernels.h) for
// batches above one trajectory per CU, the default; 3 = the pair for every batch; 1 = one wavefront per trajectory and group of parameter tiles (cp_step_poly_kernel)
inline int cp_poly_variant() { static const int v = [] { const char* e = std::getenv("PDP_CP_POLY_VARIANT"); return e ? std::atoi(e) : 2; }(); return v; }
template <class Mdl, int NT, int TPW>
int cp_step2_launch(int B, int gy, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* loss, double* grad, double* x,
                    double* u, int slice, void* st) {
    const int lds = slice * TPW * (int)sizeof(double);
    (void)hipFuncSetAttribute((const void*)cp_step_poly2_kernel<Mdl, NT, TPW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    PDP_CLEAR();
    hipLaunchKernelGGL((cp_step_poly2_kernel<Mdl, NT, TPW>), dim3((B + TPW - 1) / TPW, gy), dim3(128 * TPW), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, slice);
    return launched();
}
// MLP kernel variants (environment PDP_CP_MLP_VARIANT overrides): 2 = network in registers (pdp_cp_mlp_kernels.h), the default for networks of at
// most 4 layers of width <= 16 - four trajectories per wavefront on the 4-block MFMA for shared parameters from two trajectories per CU on, one trajectory per
// wavefront otherwise; 3 = one trajectory per wavefront for every batch (round 4's route); 4 = four per wavefront for every batch; 1 = the general adjoint kernel
// (any policy up to 8 layers x 32 units)
inline int cp_mlp_variant() { static const int v = [] { const char* e = std::getenv("PDP_CP_MLP_VARIANT"); return e ? std::atoi(e) : 2; }(); return v; }
// ---- the size-generic route of ControlPlanning.step (csrc/pdp_cp_generic_kernels.h): whatever the tuned kernels below do not take
template <class Mdl>
bool cp_policy_args_ok(const pdp_policy* pol, int p) {        // the parameter vector has the length the policy implies
    if (pol->kind == PDP_POLICY_POLY) return pol->n_pivots >= 1 && pol->n_pivots <= 16 && p == pol->n_pivots * Mdl::NU;
    if (pol->kind == PDP_POLICY_TABLE) return pol->n_basis >= 1 && pol->table != nullptr && p == pol->n_basis * Mdl::NU;
    if (pol->kind != PDP_POLICY_MLP || pol->n_layers < 1 || pol->n_layers > GEN_MAXL) return false;
    int64_t cnt = 0;
    int cols = Mdl::NX;
    for (int k = 0; k < pol->n_layers; ++k) { if (pol->sizes[k] < 1) return false; cnt += (int64_t)pol->sizes[k] * cols + pol->sizes[k]; cols = pol->sizes[k]; }
    return cnt == p && cols == Mdl::NU;
}
template <class Mdl>
bool cp_needs_generic(const pdp_policy* pol, int p) {
    if (Mdl::NX > 16 || Mdl::NU > 4) return true;                                  // beyond one tile per matrix
    if (pol->kind == PDP_POLICY_TABLE) return true;
    if (pol->kind == PDP_POLICY_MLP) {
        if (p > 512 || pol->n_layers > 8) return true;
        for (int k = 0; k < pol->n_layers; ++k) if (pol->sizes[k] > MLP_MAX_WIDTH) return true;
    }
    return false;
}
static int cp_generic_wide_bytes() {
    static const int v = [] { const char* e = std::getenv("PDP_CP_GENERIC_WIDE_BYTES"); return e ? std::atoi(e) : 96 * 1024; }();
    return v;
}
template <class Mdl>
int cp_step_generic(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* loss, double* grad, double* x, double* u,
                    void* ws, int64_t wsb, void* st) {
    if (!cp_policy_args_ok<Mdl>(pol, p)) return PDP_E_ARG;
    if (Mdl::NX > 64 || Mdl::NU > 64) return PDP_E_SIZE;          // the size-generic adjoint kernel holds one Jacobian column per lane
    const CpGenLayout L = cp_generic_layout<Mdl>(*pol, T, x != nullptr, u != nullptr, cp_generic_wide_bytes());
    if (L.rows < 1 || (size_t)L.lds_total * sizeof(double) > 160 * 1024) return PDP_E_SIZE;       // (a model whose single Jacobian row exceeds the LDS: not a policy size)
    if (L.ws_per_traj > 0 && (!ws || wsb < (int64_t)B * L.ws_per_traj * (int64_t)sizeof(double))) return PDP_E_ARG;
    const size_t lds = sizeof(double) * (size_t)L.lds_total;
    PDP_CLEAR();
    if (pol->kind == PDP_POLICY_MLP) {
        (void)hipFuncSetAttribute((const void*)cp_step_generic_kernel<Mdl, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cp_step_generic_kernel<Mdl, true>), dim3(B), dim3(64), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, L);
    } else {
        (void)hipFuncSetAttribute((const void*)cp_step_generic_kernel<Mdl, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((cp_step_generic_kernel<Mdl, false>), dim3(B), dim3(64), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, L);
    }
    return launched();
}
template <class Mdl>
int64_t cp_step_ws_bytes(int B, int T, const pdp_policy* pol, int p) {
    if constexpr (Mdl::KIND == PDP_KIND_CP) {
        if (pol && cp_needs_generic<Mdl>(pol, p)) {
            if (!cp_policy_args_ok<Mdl>(pol, p)) return 0;
            return (int64_t)B * cp_generic_layout<Mdl>(*pol, T, false, false, cp_generic_wide_bytes()).ws_per_traj * (int64_t)sizeof(double);
        }
        if (pol && pol->kind == PDP_POLICY_POLY && p <= 64 && cp_prepass(B)) return cp_prepass_ws_bytes<Mdl>(B, T);
        if (!pol || pol->kind != PDP_POLICY_MLP || pol->n_layers < 1 || pol->n_layers > 8) return 0;
        if constexpr (Mdl::NX > 16 || Mdl::NU > 4) return 0; else {
        int64_t need = 0;
        if (cp_mlp16_ok<Mdl>(*pol)) {
            need = (int64_t)B * T * 64 * (int64_t)sizeof(double);                                // register kernel: one double per lane and time step
            const int64_t n4 = cp_mlp4t_ws_doubles<Mdl>(B, T) * (int64_t)sizeof(double);          // four-trajectory kernel: activations in D layout + the trajectories
            need = n4 > need ? n4 : need;
        }
        bool offload; int rows;
        cp_adjoint_plan<Mdl>(*pol, p, T, B, device_cu_count(), true, offload, rows);
        if (offload) {
            int actw = 0;
            for (int k = 0; k + 1 < pol->n_layers; ++k) actw += pol->sizes[k];
            const int64_t a = (int64_t)B * T * actw * (int64_t)sizeof(double);
            need = a > need ? a : need;
        }
        return need;
        }
    } else { return 0; }
}
template <class Mdl>
int cp_step(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* th, int tb, double* loss, double* grad, double* x, double* u,
            void* ws, int64_t wsb, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_CP && !(Mdl::NX <= 16 && Mdl::NU <= 4)) {
        if (B <= 0 || T <= 0 || !pol || !x0 || !th || !loss || !grad) return PDP_E_ARG;
        return cp_step_generic<Mdl>(B, T, pol, p, x0, th, tb, loss, grad, x, u, ws, wsb, st);
    } else if constexpr (Mdl::KIND == PDP_KIND_CP && Mdl::NX <= 16 && Mdl::NU <= 4) {
        if (B <= 0 || T <= 0 || !pol || !x0 || !th || !loss || !grad) return PDP_E_ARG;
        if (cp_needs_generic<Mdl>(pol, p)) return cp_step_generic<Mdl>(B, T, pol, p, x0, th, tb, loss, grad, x, u, ws, wsb, st);
        if (pol->kind == PDP_POLICY_MLP || p > 64) {          // adjoint (reverse-mode) kernel: MLP policy, or many Lagrange pivots
            if (p > 512) return PDP_E_SIZE;
            if (pol->kind == PDP_POLICY_MLP) {
                int cols = Mdl::NX, cnt = 0;
                if (pol->n_layers < 1 || pol->n_layers > 8 || Mdl::NX > MLP_MAX_WIDTH) return PDP_E_SIZE;
                for (int k = 0; k < pol->n_layers; ++k) { if (pol->sizes[k] > MLP_MAX_WIDTH || pol->sizes[k] < 1) return PDP_E_SIZE; cnt += pol->sizes[k] * cols + pol->sizes[k]; cols = pol->sizes[k]; }
                if (cnt != p || cols != Mdl::NU) return PDP_E_ARG;
            } else if (p != pol->n_pivots * Mdl::NU || pol->n_pivots > 16) return PDP_E_ARG;
            // shared parameters (the reference's case: one policy for every initial state): four trajectories per wavefront on the 4-block MFMA (cp_step_mlp4t_kernel)
            // - from two trajectories per CU on (below that a wavefront per trajectory has a SIMD to itself and the same latency per step: measured 0.252 / 0.257 / 0.267 ms
            // against 0.272 for B = 64 / 256 / 512, probes/mlp4t_timing.py); PDP_CP_MLP_VARIANT=4 takes it for every batch (tests)
            if (pol->kind == PDP_POLICY_MLP && ((cp_mlp_variant() == 2 && B > 2 * device_cu_count()) || cp_mlp_variant() == 4) && tb == 0 && cp_mlp16_ok<Mdl>(*pol) &&
                ws != nullptr && wsb >= cp_mlp4t_ws_doubles<Mdl>(B, T) * (int64_t)sizeof(double)) {
                const size_t lds4 = sizeof(double) * (size_t)cp_mlp4t_layout<Mdl>().total;
                PDP_CLEAR();
                auto go4 = [&](auto kern) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
                    hipLaunchKernelGGL(kern, dim3((B + 3) / 4), dim3(64), lds4, S(st), B, T, *pol, p, x0, th, loss, grad, x, u, (double*)ws);
                };
                switch (pol->n_layers) {
                    case 1: go4(cp_step_mlp4t_kernel<Mdl, 1>); break;
                    case 2: go4(cp_step_mlp4t_kernel<Mdl, 2>); break;
                    case 3: go4(cp_step_mlp4t_kernel<Mdl, 3>); break;
                    default: go4(cp_step_mlp4t_kernel<Mdl, 4>); break;
                }
                return launched();
            }
            // PDP_CP_MLP_VARIANT=3: the one-trajectory register kernel for shared parameters as well (what round 4 ran; per-sample parameters always take it)
            if (pol->kind == PDP_POLICY_MLP && (cp_mlp_variant() >= 2) && cp_mlp16_ok<Mdl>(*pol) && ws != nullptr && wsb >= (int64_t)B * T * 64 * (int64_t)sizeof(double)) {
                // batches beyond one trajectory per SIMD: rows sized for eight workgroups per CU, i.e. two wavefronts per SIMD that fill each other's gaps (PDP_CP_MLP_LDS_KB overrides)
                static const int kb_env = [] { const char* e = std::getenv("PDP_CP_MLP_LDS_KB"); return e ? std::atoi(e) : 0; }();
                const int rows16 = cp_mlp16_rows<Mdl>(T, kb_env > 0 ? kb_env : (B > 4 * device_cu_count() ? 20 : 40));
                const size_t lds16 = sizeof(double) * (size_t)cp_mlp16_layout<Mdl>(T, rows16).total;
                if (rows16 >= 1 && lds16 <= 160 * 1024) {
                    (void)hipFuncSetAttribute((const void*)cp_step_mlp16_kernel<Mdl>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
                    PDP_CLEAR();
                    hipLaunchKernelGGL((cp_step_mlp16_kernel<Mdl>), dim3(B), dim3(64), lds16, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, rows16);
                    return launched();
                }
            }
            bool offload; int rows;
            cp_adjoint_plan<Mdl>(*pol, p, T, B, device_cu_count(), ws != nullptr && wsb >= cp_step_ws_bytes<Mdl>(B, T, pol, p), offload, rows);
            const size_t lds = sizeof(double) * (size_t)cp_adjoint_layout<Mdl>(*pol, p, T, offload, rows).total;
            if (lds > 160 * 1024) return PDP_E_SIZE;
            (void)hipFuncSetAttribute((const void*)cp_step_adjoint_kernel<Mdl>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            PDP_CLEAR();
            hipLaunchKernelGGL((cp_step_adjoint_kernel<Mdl>), dim3(B), dim3(64), lds, S(st), B, T, *pol, p, x0, th, tb, loss, grad, x, u,
                               offload ? (double*)ws : (double*)nullptr, rows);
            return launched();
        }
        if (p != pol->n_pivots * Mdl::NU || pol->n_pivots > 16) return PDP_E_ARG;
        const int nt = (p + 15) / 16;
        if (nt > 4) return PDP_E_SIZE;
        if (ws && cp_prepass(B) && wsb >= cp_prepass_ws_bytes<Mdl>(B, T)) {      // several trajectories per SIMD: rollout beforehand, one lane per trajectory
            switch (nt) {
                case 1: return cp_step_given_launch<Mdl, 1>(B, T, pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, st);
                case 2: return cp_step_given_launch<Mdl, 2>(B, T, pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, st);
                case 3: return cp_step_given_launch<Mdl, 3>(B, T, pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, st);
                default: return cp_step_given_launch<Mdl, 4>(B, T, pol, p, x0, th, tb, loss, grad, x, u, (double*)ws, st);
            }
        }
        // a batch that leaves SIMDs idle (one wavefront per trajectory, 4 SIMDs per CU) spreads the parameter tiles of a trajectory
        // over several wavefronts: each repeats the rollout and carries nt / gy of the sensitivity tiles
        int gy = (int)((int64_t)4 * device_cu_count() / B);
        gy = gy < 1 ? 1 : (gy > nt ? nt : gy);
        const int per = (nt + gy - 1) / gy;
        gy = (nt + per - 1) / per;
        // rollout wave + sensitivity wave per trajectory (pdp_cp_pair_kernels.h) once the batch exceeds one trajectory per CU; below that the one-wave kernel with its
        // parameter tiles spread over grid.y does as well (profiles/r03_pair_pipeline.txt).  PDP_CP_POLY_VARIANT=3 takes the pair for every batch (tests)
        if ((cp_poly_variant() == 2 && B > device_cu_count()) || cp_poly_variant() == 3) {
            const int cus = device_cu_count();
            const int slice = cp_pair_slice<Mdl>(T, pol->n_pivots);
            const int tpw = B <= cus ? 1 : (B <= 2 * cus ? 2 : 4);
            if ((size_t)slice * tpw * sizeof(double) <= 160 * 1024) {
                int gy2 = (int)((int64_t)2 * device_cu_count() / B);          // tiles spread over several pairs only while whole CUs would idle (measured: C4 shard, 0.090 ms against 0.111 with twice as many pairs)
                gy2 = gy2 < 1 ? 1 : (gy2 > nt ? nt : gy2);
                const int per2 = (nt + gy2 - 1) / gy2;
                gy2 = (nt + per2 - 1) / per2;
                switch (per2 * 10 + tpw) {
                    case 11: return cp_step2_launch<Mdl, 1, 1>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 12: return cp_step2_launch<Mdl, 1, 2>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 14: return cp_step2_launch<Mdl, 1, 4>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 21: return cp_step2_launch<Mdl, 2, 1>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 22: return cp_step2_launch<Mdl, 2, 2>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 24: return cp_step2_launch<Mdl, 2, 4>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 31: return cp_step2_launch<Mdl, 3, 1>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 32: return cp_step2_launch<Mdl, 3, 2>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 34: return cp_step2_launch<Mdl, 3, 4>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 41: return cp_step2_launch<Mdl, 4, 1>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    case 42: return cp_step2_launch<Mdl, 4, 2>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                    default: return cp_step2_launch<Mdl, 4, 4>(B, gy2, T, pol, p, x0, th, tb, loss, grad, x, u, slice, st);
                }
            }
        }
        switch (per) {
            case 1: return cp_step_launch<Mdl, 1>(B, gy, T, pol, p, x0, th, tb, loss, grad, x, u, st);
            case 2: return cp_step_launch<Mdl, 2>(B, gy, T, pol, p, x0, th, tb, loss, grad, x, u, st);
            case 3: return cp_step_launch<Mdl, 3>(B, gy, T, pol, p, x0, th, tb, loss, grad, x, u, st);
            default: return cp_step_launch<Mdl, 4>(B, gy, T, pol, p, x0, th, tb, loss, grad, x, u, st);
        }
    } else { return Mdl::KIND == PDP_KIND_CP ? PDP_E_SIZE : PDP_E_MODE; }
}

template <class Mdl>
int sysid_integrate(int B, int T, const double* x0, const double* u, const double* th, int tb, double* x, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_SYSID) {
        if (B <= 0 || T <= 0 || !x0 || !u || !th || !x) return PDP_E_ARG;
        PDP_CLEAR();
        hipLaunchKernelGGL((sysid_integrate_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, x0, (int)Mdl::NX, u, th, tb, x);
        return launched();
    } else { return PDP_E_MODE; }
}
template <class Mdl>
int sysid_auxsys(int B, int T, const double* x, const double* u, const double* th, int tb, double* F, double* E, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_SYSID) {
        if (B <= 0 || T <= 0 || !x || !u || !th) return PDP_E_ARG;
        const int64_t n = (int64_t)B * T;
        PDP_CLEAR();
        hipLaunchKernelGGL((sysid_auxsys_kernel<Mdl>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, S(st), B, T, x, u, th, tb, F, E);
        return launched();
    } else { return PDP_E_MODE; }
}
// Batches from which SysID.step rolls the trajectories out beforehand, one LANE per trajectory (sysid_integrate_kernel into the caller's workspace), and runs the fused
// kernel on the given trajectories: more than two trajectories per SIMD (profiles/r04_rollout_prepass.txt).  PDP_SYSID_PREPASS=0 / 1 forces it off / on.
inline bool sysid_prepass(int B) {
    static const int env = [] { const char* e = std::getenv("PDP_SYSID_PREPASS"); return e ? std::atoi(e) : -1; }();
    return env >= 0 ? env != 0 : B > 8 * device_cu_count();
}
template <class Mdl>
int64_t sysid_step_ws_bytes(int B, int T) {
    if constexpr (Mdl::KIND == PDP_KIND_SYSID) return sysid_prepass(B) ? (int64_t)B * (T + 1) * Mdl::NX * (int64_t)sizeof(double) : 0;
    else return 0;
}
template <class Mdl>
int sysid_step(int B, int T, const double* u, const double* xobs, const double* th, int tb, double* loss, double* grad, void* ws, int64_t wsb, void* st) {
    if constexpr (Mdl::KIND == PDP_KIND_SYSID && Mdl::NX <= 16 && Mdl::NP <= 64) {
        if (B <= 0 || T <= 0 || !u || !xobs || !th || !loss || !grad) return PDP_E_ARG;
        const double* xgiven = nullptr;
        if (ws && sysid_prepass(B)) {           // (no workspace: the kernel rolls out itself, whatever the batch)
            if (wsb < sysid_step_ws_bytes<Mdl>(B, T)) return PDP_E_ARG;
            PDP_CLEAR();
            hipLaunchKernelGGL((sysid_integrate_kernel<Mdl>), dim3((B + 63) / 64), dim3(64), 0, S(st), B, T, xobs, (int)((T + 1) * Mdl::NX), u, th, tb, (double*)ws);
            if (const int rc = launched(); rc != 0) return rc;
            xgiven = (const double*)ws;
        }
        constexpr int NT = (Mdl::NP + 15) / 16;
        static const int rows_env = [] { const char* e = std::getenv("PDP_SYSID_ROWS"); return e ? std::atoi(e) : 0; }();
        static const int wgs_env = [] { const char* e = std::getenv("PDP_SYSID_GIVEN_WGS"); return e ? std::atoi(e) : 0; }();
        const int rows = rows_env > 0 ? (rows_env < Mdl::CHUNK ? rows_env : Mdl::CHUNK)
                         : (xgiven ? sysid_rows_given<Mdl>(T, wgs_env > 0 ? wgs_env : 12) : sysid_rows<Mdl>(B, T, device_cu_count()));
        const size_t lds = sizeof(double) * (size_t)sysid_slice<Mdl>(T, rows, xgiven != nullptr);
        if (lds > 150 * 1024) return PDP_E_SIZE;
        // PDP_SYSID_VARIANT: 2 = rollout wave + sensitivity wave per trajectory (pdp_cp_pair_kernels.h), the default; 1 = one wavefront per trajectory
        static const int variant = [] { const char* e = std::getenv("PDP_SYSID_VARIANT"); return e ? std::atoi(e) : 2; }();
        // the pair pays while SIMDs would idle (B = 256, T = 200: 0.105 -> 0.072 ms); once every SIMD has a trajectory the two waves only share what one had
        // (B = 1024: 0.0667 against 0.0685 ms, profiles/r03_pair_pipeline.txt) - the one-wave kernel stays for those batches
        if (variant == 2 && B <= 2 * device_cu_count() && !xgiven) {
            const int cus = device_cu_count(), slice = sysid_slice<Mdl>(T);
            const int tpw = B <= cus ? 1 : 2;
            const int lds2 = slice * tpw * (int)sizeof(double);
            if (lds2 <= 160 * 1024) {
                PDP_CLEAR();
                if (tpw == 1) { (void)hipFuncSetAttribute((const void*)sysid_step2_kernel<Mdl, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
                                hipLaunchKernelGGL((sysid_step2_kernel<Mdl, NT, 1>), dim3(B), dim3(128), lds2, S(st), B, T, u, xobs, th, tb, loss, grad, slice); }
                else { (void)hipFuncSetAttribute((const void*)sysid_step2_kernel<Mdl, NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
                       hipLaunchKernelGGL((sysid_step2_kernel<Mdl, NT, 2>), dim3((B + 1) / 2), dim3(256), lds2, S(st), B, T, u, xobs, th, tb, loss, grad, slice); }
                return launched();
            }
        }
        PDP_CLEAR();
        if (xgiven) {
            (void)hipFuncSetAttribute((const void*)sysid_step_kernel<Mdl, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((sysid_step_kernel<Mdl, NT, true>), dim3(B), dim3(64), lds, S(st), B, T, u, xobs, th, tb, loss, grad, rows, xgiven);
        } else {
            (void)hipFuncSetAttribute((const void*)sysid_step_kernel<Mdl, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((sysid_step_kernel<Mdl, NT, false>), dim3(B), dim3(64), lds, S(st), B, T, u, xobs, th, tb, loss, grad, rows, xgiven);
        }
        return launched();
    } else { return Mdl::KIND == PDP_KIND_SYSID ? PDP_E_SIZE : PDP_E_MODE; }
}

template <class Mdl> int nnz_path() { return Mdl::PATH_NVAR; }

}  // namespace

extern "C" {

void pdp_model_get_info(pdp_model_info* info) {
    if (!info) return;
    info->kind = PdpModel::KIND; info->n = PdpModel::NX; info->m = PdpModel::NU; info->p = PdpModel::NP;
    info->nnz_path = nnz_path<PdpModel>(); info->chunk = PdpModel::CHUNK; info->name = PdpModel::NAME;
}
int pdp_oc_rollout_batched(int B, int T, const double* x0, const double* u, const double* theta, int tb, double* x, double* cost, void* stream) {
    return oc_rollout<PdpModel>(B, T, x0, u, theta, tb, x, cost, stream);
}
int pdp_oc_rollout_feedback_batched(int B, int T, const double* x0, const double* ubar, const double* xbar, const double* gains, const double* alpha,
                                    const double* theta, int tb, double* x, double* u, double* cost, void* stream) {
    return oc_rollout_fb<PdpModel>(B, T, x0, ubar, xbar, gains, alpha, theta, tb, x, u, cost, stream);
}
int pdp_oc_ms_residuals_batched(int B, int T, const double* x, const double* u, const double* lam, const double* theta, int tb, double* c, double* rx,
                                double* ru, double* cost, void* stream) {
    return oc_ms_residuals<PdpModel>(B, T, x, u, lam, theta, tb, c, rx, ru, cost, stream);
}
int pdp_oc_costate_batched(int B, int T, const double* x, const double* u, const double* theta, int tb, double* lam, void* stream) {
    return oc_costate<PdpModel>(B, T, x, u, theta, tb, lam, stream);
}
int pdp_oc_auxsys_batched(int B, int T, const double* x, const double* u, const double* lam, const double* theta, int tb, const pdp_oc_auxsys* out,
                          void* stream) {
    return oc_auxsys<PdpModel>(B, T, x, u, lam, theta, tb, out, stream);
}
int64_t pdp_oc_solve_workspace_bytes(int B, int T, int ls_trials) {
    if constexpr (PdpModel::KIND == PDP_KIND_OC) return OcSolveWs<PdpModel>(nullptr, B, T, ls_trials > 0 ? ls_trials : 10).bytes; else return 0;
}
int pdp_oc_solve_batched(int B, int T, const double* x0, const double* theta, int tb, double* u, double* x, double* lam, double* cost,
                         double* grad_norm, int32_t* converged, double* gains, const pdp_oc_solve_opts* opts, int* iterations, void* workspace,
                         int64_t workspace_bytes, void* stream) {
    return oc_solve<PdpModel>(B, T, x0, theta, tb, u, x, lam, cost, grad_norm, converged, gains, opts, iterations, workspace, workspace_bytes, stream);
}
int64_t pdp_oc_solve_ms_workspace_bytes(int B, int T, int max_iter) { return oc_solve_ms_ws_bytes<PdpModel>(B, T, max_iter); }
int pdp_oc_solve_ms_batched(int B, int T, const double* x0, const double* theta, int tb, double* x, double* u, double* lam, double* cost,
                            double* resid, int32_t* converged, int32_t* iterations, int32_t* status, double* gains, double* iter_log,
                            const pdp_oc_ms_opts* opts, void* workspace, int64_t workspace_bytes, void* stream) {
    return oc_solve_ms<PdpModel>(B, T, x0, theta, tb, x, u, lam, cost, resid, converged, iterations, status, gains, iter_log, opts, workspace, workspace_bytes,
                                 stream);
}
int64_t pdp_oc_pdp_workspace_bytes(int B, int T) {
    if constexpr (PdpModel::KIND == PDP_KIND_OC) return oc_ws_bytes<PdpModel>(B, T); else return 0;
}
int pdp_oc_pdp_grad_batched(int B, int T, int flags, const double* x0, const double* u, const double* theta, int tb, const double* demo_x,
                            const double* demo_u, double* x, double* lam, double* loss, double* grad, double* dxdp, double* dudp, int32_t* status,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    return oc_pdp<PdpModel>(B, T, flags, x0, u, theta, tb, demo_x, demo_u, x, lam, loss, grad, dxdp, dudp, nullptr, nullptr, status, workspace, workspace_bytes, stream);
}
int64_t pdp_oc_riccati_doubles(void) {
    if constexpr (PdpModel::KIND == PDP_KIND_OC) return oc_riccati_doubles<PdpModel>(); else return 0;
}
int64_t pdp_oc_predict_record_floats(void) {
    if constexpr (PdpModel::KIND == PDP_KIND_OC) return PredRec<PdpModel>::SIZE; else return 0;
}
int pdp_oc_pdp_grad_sens_batched(int B, int T, int flags, const double* x0, const double* u, const double* theta, int tb, const double* demo_x,
                                 const double* demo_u, double* x, double* lam, double* loss, double* grad, const pdp_oc_sens_out* sens,
                                 int32_t* status, void* workspace, int64_t workspace_bytes, void* stream) {
    const pdp_oc_sens_out none = {nullptr, nullptr, nullptr, nullptr};
    const pdp_oc_sens_out& so = sens ? *sens : none;
    return oc_pdp<PdpModel>(B, T, flags, x0, u, theta, tb, demo_x, demo_u, x, lam, loss, grad, so.dxdp, so.dudp, so.riccati, so.predict_record, status, workspace,
                            workspace_bytes, stream);
}
int pdp_oc_predict_record_batched(int B, int T, const double* dtheta, int dtheta_bstride, const float* predict_record, double* x, double* u, double* lam,
                                  void* stream) {
    return oc_predict_rec<PdpModel>(B, T, dtheta, dtheta_bstride, predict_record, x, u, lam, stream);
}
int pdp_oc_predict_batched(int B, int T, const double* dtheta, int dtheta_bstride, const double* dxdp, const double* dudp, const double* riccati, double* x,
                           double* u, double* lam, void* stream) {
    return oc_predict<PdpModel>(B, T, dtheta, dtheta_bstride, dxdp, dudp, riccati, x, u, lam, stream);
}
int pdp_cp_integrate_batched(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* theta, int tb, double* x, double* u,
                             double* cost, void* stream) {
    return cp_integrate<PdpModel>(B, T, pol, p, x0, theta, tb, x, u, cost, stream);
}
int pdp_cp_auxsys_batched(int B, int T, const pdp_policy* pol, int p, const double* x, const double* u, const double* theta, int tb, double* dynF,
                          double* dynG, double* dUx, double* dUe, double* dcx, double* dcu, double* dhx, void* stream) {
    return cp_auxsys<PdpModel>(B, T, pol, p, x, u, theta, tb, dynF, dynG, dUx, dUe, dcx, dcu, dhx, stream);
}
int64_t pdp_cp_step_workspace_bytes(int B, int T, const pdp_policy* pol, int p) { return cp_step_ws_bytes<PdpModel>(B, T, pol, p); }
int pdp_cp_step_batched(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* theta, int tb, double* loss, double* grad,
                        double* x, double* u, void* workspace, int64_t workspace_bytes, void* stream) {
    return cp_step<PdpModel>(B, T, pol, p, x0, theta, tb, loss, grad, x, u, workspace, workspace_bytes, stream);
}
int pdp_sysid_integrate_batched(int B, int T, const double* x0, const double* u, const double* theta, int tb, double* x, void* stream) {
    return sysid_integrate<PdpModel>(B, T, x0, u, theta, tb, x, stream);
}
int pdp_sysid_auxsys_batched(int B, int T, const double* x, const double* u, const double* theta, int tb, double* dynF, double* dynE, void* stream) {
    return sysid_auxsys<PdpModel>(B, T, x, u, theta, tb, dynF, dynE, stream);
}
int pdp_sysid_step_batched(int B, int T, const double* u, const double* x_obs, const double* theta, int tb, double* loss, double* grad, void* stream) {
    return sysid_step<PdpModel>(B, T, u, x_obs, theta, tb, loss, grad, nullptr, 0, stream);
}
int64_t pdp_sysid_step_workspace_bytes(int B, int T) { return sysid_step_ws_bytes<PdpModel>(B, T); }
int pdp_sysid_step_ws_batched(int B, int T, const double* u, const double* x_obs, const double* theta, int tb, double* loss, double* grad, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    return sysid_step<PdpModel>(B, T, u, x_obs, theta, tb, loss, grad, workspace, workspace_bytes, stream);
}

}  // extern "C"
