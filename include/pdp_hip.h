/* pdp_hip.h - C-ABI of the MI355X-native batched PDP inner loop.
 *
 * The reference (wanxinjin/Pontryagin-Differentiable-Programming) has no FFI seam: its boundary is the
 * Python class surface of PDP/PDP.py and, beneath it, casadi.Function.__call__ + numpy.linalg.  This
 * header is the C boundary inserted beneath that class surface.  Every entry point
 * cites the reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - all arrays are contiguous IEEE fp64, row-major per matrix, batch-major: [B][T][rows][cols];
 *     all pointers are DEVICE pointers (HBM); the caller owns every buffer; the library allocates nothing
 *     persistent and keeps no global state besides immutable generated model code;
 *   - `stream` is a hipStream_t passed as void*; work is stream-ordered, nothing synchronises the device;
 *   - return value: 0 ok, <0 argument error (PDP_E_*); numerical trouble is reported per trajectory in
 *     `status[B]` (bit 0: non-finite value met, bit 1: pivot below 1e-300 in the m x m solve), the analogue
 *     of numpy.linalg.LinAlgError in the reference (PDP.py:566, 575);
 *   - a matrix family that is time-invariant may be passed with time stride 0, one shared by the whole
 *     batch with batch stride 0 (strides in doubles, see pdp_lqr_problem).
 *
 * Two libraries export these symbols:
 *   libpdp_hip.so              model-independent kernels  (section A)
 *   libpdp_model_<name>.so     one per generated model    (section B), built from generated HIP source
 */
#ifndef PDP_HIP_H
#define PDP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDP_E_ARG (-1)      /* null pointer / non-positive size */
#define PDP_E_SIZE (-2)     /* dimension outside what the kernels are built for (see each entry point) */
#define PDP_E_LAUNCH (-3)   /* HIP launch error (hipGetLastError() has the detail) */
#define PDP_E_MODE (-4)     /* entry point not provided by this model kind */

#define PDP_STATUS_NONFINITE 1
#define PDP_STATUS_PIVOT 2
#define PDP_STATUS_INDEFINITE 256 /* pdp_lqr_solve_batched, generic kernel (n > 16 or m > 4) only: some Quu_t = Huu + G'PG of the sweep is not positive definite - the result is
                                     the stationary point of the LQ problem, not its minimiser (the inertia test of the multiple-shooting OC route for large systems) */

/* library identification: returns a static string "pdp_hip <version> gfx950" */
const char* pdp_hip_version(void);

/* ------------------------------------------------------------------------------------------------------
 * Section A - model-independent kernels (libpdp_hip.so)
 * ------------------------------------------------------------------------------------------------------ */

/* One matrix family of the auxiliary control system: base pointer + strides in doubles.
 * element(b,t,i,j) = ptr[b*bstride + t*tstride + i*cols + j]; ptr may be NULL where noted (treated as 0). */
typedef struct pdp_mat {
    const double* ptr;
    int64_t bstride;
    int64_t tstride;
} pdp_mat;

/* Inputs of LQR.lqrSolver (PDP/PDP.py:446-555 normalises them; 557-608 consumes them).
 * Hux is accepted by the reference but never read - Hxu^T is used (PDP.py:569, 593, 598) - so it is absent. */
typedef struct pdp_lqr_problem {
    int B, T, n, m, p;
    pdp_mat F;    /* dynF  [n x n] */
    pdp_mat G;    /* dynG  [n x m] */
    pdp_mat E;    /* dynE  [n x p]  (NULL = 0, PDP.py:496) */
    pdp_mat Hxx;  /* [n x n] */
    pdp_mat Hxu;  /* [n x m]        (NULL = 0, PDP.py:517-518) */
    pdp_mat Hxe;  /* [n x p]        (NULL = 0, PDP.py:537-538) */
    pdp_mat Huu;  /* [m x m] */
    pdp_mat Hue;  /* [m x p]        (NULL = 0, PDP.py:547-548) */
    pdp_mat hxx;  /* terminal [n x n], tstride ignored */
    pdp_mat hxe;  /* terminal [n x p], tstride ignored (the reference requires it, PDP.py:562) */
    pdp_mat X0;   /* ini_state [n x p] (NULL = 0), tstride ignored */
} pdp_lqr_problem;

/* Size in bytes of the scratch buffer pdp_lqr_solve_batched needs (feedback gains, and P/W when Lam != NULL). */
int64_t pdp_lqr_workspace_bytes(int B, int T, int n, int m, int p, int want_costate);

/* Batched LQR.lqrSolver (PDP/PDP.py:446-615): backward Riccati sweep (557-580) and forward rollout of
 * X [B][T+1][n][p], U [B][T][m][p], Lam [B][T][n][p] (582-608; Lam may be NULL).  One wavefront per
 * trajectory.  n <= 16, m <= 4, p <= 64 - m: 16x16 fp64 MFMA tiles held in registers; n <= 4 and m + p <= 16: FOUR trajectories per
 * wavefront, block-diagonal in the tile, every product on the 4-block MFMA v_mfma_f64_4x4x4 (pdp_riccati_small.h); 16 < n <= 32 or
 * 4 < m <= 8 with p <= 32: a generic LDS kernel (plain fp64 loops).  Beyond those limits PDP_E_SIZE; more parameter columns than one
 * launch carries are solved in column blocks by the caller (the columns are independent given the gains; runtime.lqr_solve does it). */
int pdp_lqr_solve_batched(const pdp_lqr_problem* prob, double* X, double* U, double* Lam, int32_t* status,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* Batched ControlPlanning.integrateAuxSys (PDP/PDP.py:813-838): U_t = Ux_t X_t + Ue_t, X_{t+1} = F_t X_t + G_t U_t.
 * F [B][T][n][n], G [B][T][n][m], Ux [B][T][m][n], Ue [B][T][m][p], X0 [B][n][p] (NULL = 0)
 * -> X [B][T+1][n][p], U [B][T][m][p]. */
int pdp_cp_aux_integrate_batched(int B, int T, int n, int m, int p, const double* F, const double* G, const double* Ux,
                                 const double* Ue, const double* X0, double* X, double* U, void* stream);

/* Chain rule of ControlPlanning.step (PDP/PDP.py:869-876): grad[b][j] = sum_t dcx[b][t][:] X[b][t][:][j] + dcu[b][t][:] U[b][t][:][j]
 * + dhx[b][:] X[b][T][:][j].  dcx [B][T][n], dcu [B][T][m], dhx [B][n], X [B][T+1][n][p], U [B][T][m][p] -> grad [B][p]. */
int pdp_cp_grad_contract_batched(int B, int T, int n, int m, int p, const double* dcx, const double* dcu, const double* dhx,
                                 const double* X, const double* U, double* grad, void* stream);

/* The parameter update of the reference's gradient-descent loops as ONE launch (Examples/IRL/cartpole/cartpole_PDP.py:76-80: loss and dp accumulated over the
 * demonstrations, current_parameter -= lr * dp; PDP.py:1293-1294: batch means): mean of loss [B] and of grad [B][grad_bstride >= p] over the batch in a fixed summation order,
 *     dtheta = -lr * mean gradient,  theta += dtheta,  loss_trace[k] = mean loss,  parameter_trace[k][:] = theta      (k = counters[0]; traces may be NULL, rows >= trace_len are dropped)
 * and the counters of a loop that never synchronises with the host: counters[0] += 1 (iterations done), counters[1] += #(converged[b] == 0), counters[2] += #(status[b] != 0),
 * counters[3] += sum iterations[b]   (status / converged / iterations: the int32 outputs of the gradient unit and of pdp_oc_solve_ms_batched, each may be NULL).
 * p <= 1023.  A loop iteration then is three launches - solve, gradient unit, this - and is what pdp_amd.irl.IRLLoop records as a hipGraph. */
int pdp_gd_update_batched(int B, int p, const double* loss, const double* grad, int grad_bstride, const int32_t* status, const int32_t* converged,
                          const int32_t* iterations, double lr, double* theta, double* dtheta, double* loss_trace, double* parameter_trace,
                          int64_t trace_len, int64_t* counters, void* stream);

/* Batched SysID.integrateAuxSys (PDP/PDP.py:1241-1259): X_{t+1} = F_t X_t + E_t.
 * F [B][T][n][n], E [B][T][n][p], X0 [B][n][p] (NULL = 0) -> X [B][T+1][n][p]. */
int pdp_sysid_aux_integrate_batched(int B, int T, int n, int p, const double* F, const double* E, const double* X0,
                                    double* X, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Section B - per-model entry points (libpdp_model_<name>.so, generated by pdp_amd.codegen from the
 * symbolic problem definition; the CasADi Function objects of diffPMP, PDP.py:222-270, become device code)
 * ------------------------------------------------------------------------------------------------------ */

#define PDP_KIND_OC 0      /* OCSys            (PDP.py:57-314)    x+ = f(x,u,theta), c(x,u,theta), h(x,theta) */
#define PDP_KIND_CP 1      /* ControlPlanning  (PDP.py:640-878)   x+ = f(x,u), c(x,u), h(x), policy u = pi(t,x,theta) */
#define PDP_KIND_SYSID 2   /* SysID            (PDP.py:1157-1296) x+ = f(x,u,theta) */

typedef struct pdp_model_info {
    int kind;          /* PDP_KIND_* */
    int n, m, p;       /* state, control, auxvar dimension (p = 0 for PDP_KIND_CP: theta lives in the policy) */
    int nnz_path;      /* structural non-zeros of the per-step aux matrices (packed LDS pool width) */
    int chunk;         /* time steps evaluated per lane-parallel aux pass of the fused kernel */
    const char* name;  /* "<model>_<kind>_<hash>" */
} pdp_model_info;

void pdp_model_get_info(pdp_model_info* info);

/* theta handling, all per-model entry points: theta[(b * theta_bstride) + k]; theta_bstride = 0 shares one
 * parameter vector over the batch (reference semantics), = p gives every trajectory its own. */

/* ---- PDP_KIND_OC ----------------------------------------------------------------------------------- */

/* Forward trajectory for given controls: x_{t+1} = f(x_t,u_t,theta) (the NLP equality constraints of
 * OCSys.ocSolver, PDP.py:148-170) and J = sum c + h (157-173).
 * x0 [B][n], u [B][T][m] -> x [B][T+1][n], cost [B] (cost may be NULL). */
int pdp_oc_rollout_batched(int B, int T, const double* x0, const double* u, const double* theta, int theta_bstride,
                           double* x, double* cost, void* stream);

/* Closed-loop rollout used by the batched OC solver that stands where OCSys.ocSolver calls IPOPT (PDP.py:121-220):
 *   u_t = ubar_t - alpha_b k_t - K_t (x_t - xbar_t),  x_{t+1} = f(x_t,u_t,theta),  cost = sum c + h
 * with the feedback gains in the layout pdp_lqr_solve_batched leaves in its workspace for p = 1:
 * gains[b][t] = { K^T [n][m], k [m] }.  x0 [B][n], ubar [B][T][m], xbar [B][T+1][n], alpha [B] -> x [B][T+1][n], u [B][T][m], cost [B]. */
int pdp_oc_rollout_feedback_batched(int B, int T, const double* x0, const double* ubar, const double* xbar, const double* gains,
                                    const double* alpha, const double* theta, int theta_bstride, double* x, double* u, double* cost,
                                    void* stream);

/* Residuals of ocSolver's multiple-shooting NLP (PDP.py:131-182) at a point (x, u, lam), lam[t] = multiplier of f(x_t,u_t) - x_{t+1}: defects c [B][T][n],
 * Lagrangian gradients rx [B][T+1][n] (node 0: 0 - x_0 is fixed; 0 < t < T: H_x(x_t,u_t,lam_t) - lam_{t-1}; T: h_x(x_T) - lam_{T-1}) and ru [B][T][m] = H_u,
 * stage costs cost [B][T+1] (final cost at T).  Size-generic (any n, m): with pdp_oc_auxsys_batched and pdp_lqr_solve_batched the building block of the
 * kernel-by-kernel multiple-shooting route for problems beyond the solver kernel's n <= 16, m <= 4 (ocsolver.solve_batch_ms_generic). */
int pdp_oc_ms_residuals_batched(int B, int T, const double* x, const double* u, const double* lam, const double* theta, int theta_bstride,
                                double* c, double* rx, double* ru, double* cost, void* stream);

/* PMP costate recursion, ocSolver costate_option=1 (PDP.py:199-209): lam[T-1] = h_x(x_T),
 * lam[k-1] = c_x(x_k,u_k) + f_x(x_k,u_k)^T lam[k].  lam [B][T][n] with lam[t] = lambda_{t+1}. */
int pdp_oc_costate_batched(int B, int T, const double* x, const double* u, const double* theta, int theta_bstride,
                           double* lam, void* stream);

/* OCSys.getAuxSys (PDP.py:272-314), materialised: dynF [B][T][n][n], dynG [B][T][n][m], dynE [B][T][n][p],
 * Hxx, Hxu [n x m], Hxe [n x p], Hux [m x n], Huu [m x m], Hue [m x p] (all [B][T]...), hxx [B][n][n], hxe [B][n][p].
 * Any output pointer may be NULL (skipped). */
typedef struct pdp_oc_auxsys {
    double *dynF, *dynG, *dynE, *Hxx, *Hxu, *Hxe, *Hux, *Huu, *Hue, *hxx, *hxe;
    double* dHu; /* optional extra (not part of the reference dict): dH/du [B][T][m] = c_u + f_u' lambda_{t+1}, the PMP
                    stationarity residual (dHu_fn, PDP.py:245-246) used by the batched OC solver */
    const double* Huu_damp; /* optional INPUT [B]: Huu[b][t] += Huu_damp[b] I (Levenberg-Marquardt damping of the OC solver) */
} pdp_oc_auxsys;
int pdp_oc_auxsys_batched(int B, int T, const double* x, const double* u, const double* lam, const double* theta,
                          int theta_bstride, const pdp_oc_auxsys* out, void* stream);

/* Batched optimal-control solve standing where OCSys.ocSolver hands its multiple-shooting NLP to IPOPT (PDP.py:121-220):
 * stagewise Newton (Gauss-Newton + Levenberg-Marquardt damping far from the optimum) on the single-shooting problem, every
 * trajectory of the batch in parallel, the iterations sequenced by the library on `stream` with no host round trip per
 * iteration - costates -> H_u, Hessians (this library's kernels) -> LQ sub-problem (the LQR.lqrSolver kernel, p = 1) ->
 * closed-loop line search over ls_trials step lengths -> per-sample acceptance.  The count of converged samples is polled
 * every `check_every` iterations (that poll synchronises the stream).  Convergence: |H_u|_inf <= tol (1 + |u|_inf); at
 * convergence (x, u, lam) satisfy the KKT conditions IPOPT solves, lam[t] = lambda_{t+1} = its `lam_g`.
 *   in : x0 [B][n], theta, u [B][T][m] (initial controls, overwritten by the solution)
 *   out: x [B][T+1][n], lam [B][T][n], cost [B], grad_norm [B], converged [B], *iterations (host),
 *        gains [B][T][n m + m] (optional, NULL = skip): the LQR feedback {K^T, k} around the solution (closed-loop warm starts)
 * straggler_patience > 0: stop once >= 90 % of the batch has converged and no further sample converged for that many
 * iterations (the caller re-solves the rest from a neighbour's solution).  n <= 16, m <= 4. */
typedef struct pdp_oc_solve_opts {
    double tol, newton_switch;
    int max_iter, check_every, ls_trials, straggler_patience, print_level;
} pdp_oc_solve_opts;
int64_t pdp_oc_solve_workspace_bytes(int B, int T, int ls_trials);
int pdp_oc_solve_batched(int B, int T, const double* x0, const double* theta, int theta_bstride, double* u, double* x, double* lam,
                         double* cost, double* grad_norm, int32_t* converged, double* gains, const pdp_oc_solve_opts* opts,
                         int* iterations, void* workspace, int64_t workspace_bytes, void* stream);

/* OCSys.ocSolver (PDP.py:121-220) as the reference poses it: the multiple-shooting NLP
 *     min sum_t c(x_t,u_t) + h(x_T)  over x_1..x_T, u_0..u_{T-1}   s.t.  f(x_t,u_t) - x_{t+1} = 0,  x_0 = ini_state   (PDP.py:131-179)
 * solved from the reference's all-zero initial guess (PDP.py:155,166) by IPOPT's algorithm for the equality-constrained case
 * (Waechter & Biegler 2006: primal-dual Newton step, inertia correction, filter line search (second-order correction: optional, see below), least-squares
 * initial multipliers; CPU restatement: oracle/ipopt_ms.py).  A persistent pair of wavefronts per trajectory (runner: IPOPT's control flow and the Riccati /
 * forward chains on the MFMA tiles; evaluator: KKT matrices, trial-point residuals, multiplier step with one lane per stage - csrc/pdp_ocsolve2_kernels.h;
 * the one-wavefront kernel of round 2, csrc/pdp_ocsolve_kernels.h, stays behind PDP_MS_VARIANT=1) runs ALL iterations inside one launch: the
 * Newton step is an LQ problem in homogeneous form on the Riccati tiles, trial points are evaluated with one lane per stage; no host round
 * trip and no batch-wide synchronisation.  Finite bounds on states / controls are not part of this entry point: the Python layer
 * (ocsolver.solve_batch_bounded) wraps it in a log-barrier continuation.
 *   in    : x0 [B][n], theta;   with PDP_MS_WARM also x, u, lam (starting point, e.g. the solution at a neighbouring theta)
 *   in/out: x [B][T+1][n], u [B][T][m], lam [B][T][n]  (lam[t] = multiplier of f(x_t,u_t) - x_{t+1} = IPOPT's lam_g = costate_traj_opt[t])
 *   out   : cost [B], resid [B][2] (max |defect|, max |grad Lagrangian|), converged [B], iterations [B], status [B], optional
 *           gains [B][T][n m + m] ({K^T, k} of the last Newton step, layout of pdp_oc_rollout_feedback_batched); any may be NULL.
 * Convergence: max|defect| <= tol (1 + max|x|,|u|) and max|grad L| <= tol (1 + max|lam|).  status bits: PDP_STATUS_NONFINITE,
 * PDP_MS_RESTORATION (the line search fell below alpha_min and no restoration was possible - see below: fall back to
 * pdp_oc_solve_batched), PDP_MS_RESTORED (informational: a restoration took place), PDP_MS_MAXITER, PDP_MS_INERTIA (no positive definite reduced Hessian up to dw = 1e20), PDP_MS_NOGAINS (gains were
 * requested but the returned point has no complete positive definite sweep - not converged: zeros are written), PDP_MS_INTERNAL (the hand-over
 * between the two wavefronts timed out: a bug, never expected; the trajectory is returned unconverged instead of hanging).  n <= 16, m <= 4.
 * Restoration (round 3, runner / evaluator kernel only).  Where IPOPT's line search switches to its feasibility restoration phase (alpha < alpha_min), the
 * kernel adds the current point to the filter, keeps the controls and replaces the states by their rollout from x0 (constraint violation 0, acceptable to
 * every filter entry - what the filter method requires of a restoration phase; IPOPT's own restoration NLP is not restated), resets the multipliers to the
 * least-squares estimate as IPOPT does after a restoration, and continues; the iteration counts as one.  Not possible at a feasible point or when the rollout
 * overflows: PDP_MS_RESTORATION as before.  opts.flags & PDP_MS_NO_RESTORATION switches it off.
 * Second-order correction (round 5, runner / evaluator kernel only; opts.flags & PDP_MS_WITH_SOC; IPOPT's max_soc = 4, kappa_soc = 0.99).  When the first trial point of
 * an iteration is rejected and its constraint violation is not below the iterate's, up to four corrected steps (same KKT matrix, constraint block
 * alpha c(x_k) + c(x_k + alpha d), accumulated) are tried before the step is halved; status gets PDP_MS_SOC where one was taken, and such an iteration shows MINUS its
 * test step length in the iteration log's alpha column.  Each correction costs a Newton sweep here (IPOPT: a back-substitution).  OFF by default: every stored optimum
 * is reached either way (cart-pole demo 0 in 35 iterations instead of 45), but on the reference's own cart-pole IRL run (stored trace, first row, demonstration 4 - a
 * non-convex solve with inertia corrections at most iterations) the iteration WITH the published correction ends in another stationary point (cost 1513.67 after 2490
 * iterations) where the reference's IPOPT run and the iteration without it end in 623.79: as a model of what the reference's solver returned, the restatement is
 * better without. */
#define PDP_MS_WARM 1
#define PDP_MS_NO_RESTORATION 2   /* opts.flags: return PDP_MS_RESTORATION instead of restoring (the behaviour before round 3) */
#define PDP_MS_FROM_CONTROLS 8    /* opts.flags, with PDP_MS_WARM (runner / evaluator kernel): start from the controls in u only - x becomes their rollout from x0,
                                     lam the least-squares multiplier estimate; the contents of x and lam on entry are ignored */
#define PDP_MS_RESTORATION 4
#define PDP_MS_MAXITER 8
#define PDP_MS_INERTIA 16
#define PDP_MS_NOGAINS 32
#define PDP_MS_INTERNAL 64
#define PDP_MS_RESTORED 128       /* informational: the line search fell below alpha_min at least once and the feasibility restoration below was used */
#define PDP_MS_PREDICT 16         /* opts.flags, with PDP_MS_WARM: the starting point is the FIRST-ORDER PREDICTION from (x, u, lam) on entry - the solution at the previous
                                     parameter - for the step opts.dtheta:  x + X dtheta, u + U dtheta, lam_t + P_{t+1} X_{t+1} dtheta + W_{t+1} dtheta  with the outputs
                                     dxdp, dudp, riccati of pdp_oc_pdp_grad_sens_batched at that solution (what pdp_oc_predict_batched computes, applied while the kernel
                                     loads the point: no extra launch, no copy).  riccati may be NULL: multipliers as they are */
#define PDP_MS_PREDICT_PRIMAL 32  /* opts.flags, with PDP_MS_PREDICT and opts.predict_record: states and controls only - the multipliers stay as they are and the P | W part
                                     of the record is not read (it need not have been written: PDP_OC_RECORD_PRIMAL) */
#define PDP_MS_PREDICT_GUARD 64   /* opts.flags, with PDP_MS_PREDICT (runner / evaluator kernel): the previous solution itself (the plain warm start) is evaluated beside its
                                     prediction and the prediction is kept only if its scaled KKT error - max(inf_pr / (1 + max|x|,|u|), inf_du / (1 + max|lam|)), the
                                     quantities of the convergence test - is finite and not larger; otherwise the solve starts from the previous solution and status gets
                                     PDP_MS_PREDICT_REJECTED.  One more residual pass per solve.  Needed where parameter steps are large against the curvature: on the
                                     reference's stored rocket IRL run the unguarded prediction of row 1 sends Newton's method to another stationary point */
#define PDP_MS_GUARD_TRUST 0.02     /* PDP_MS_PREDICT_GUARD: a prediction that changes no state or control by more than this fraction of max(1, |its value|) is kept without the check */
#define PDP_MS_PREDICT_REJECTED 512 /* status, informational: PDP_MS_PREDICT_GUARD preferred the previous solution to its prediction */
#define PDP_MS_WITH_SOC 128         /* opts.flags: second-order correction in the line search (see above; off by default) */
#define PDP_MS_SOC 1024             /* status, informational: at least one iteration accepted a second-order-corrected step */
#define PDP_MS_WITH_WATCHDOG 256    /* opts.flags (runner / evaluator kernel; with PDP_MS_WITH_SOC beside it: IPOPT's default pair, as restated): IPOPT's watchdog in the line search - after 10 consecutive iterations
                                     * with a shortened step the iterate and its direction are stored and up to 3 full steps are taken whether acceptable or not, each judged from the
                                     * stored point; the first acceptable one ends the procedure, otherwise the stored iterate comes back and is searched from alpha = 1/2.  IPOPT has it
                                     * on by default (watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3) and the reference takes IPOPT's defaults (PDP/PDP.py:178-182);
                                     * here it is OPT-IN because it is restated from the structure of IPOPT's line search without a run of IPOPT to pin it on (oracle/ipopt_ms.py:
                                     * solve(watchdog=True) is the checker).  What it is for: cold solves that crawl - rocket, T = 100, zero guess: 93 % of the iterations of the solves
                                     * that miss the 300-iteration limit accept steps of 1e-3 (profiles/r06_solver_iterlog_stats.txt, r06_watchdog_experiment.txt).  Every stored
                                     * optimum of the reference is reached without it. */
#define PDP_MS_WATCHDOG 2048        /* status, informational: at least one watchdog procedure was started */
typedef struct pdp_oc_ms_opts {
    double tol;
    int max_iter;
    int flags;    /* PDP_MS_WARM, PDP_MS_NO_RESTORATION, PDP_MS_FROM_CONTROLS, PDP_MS_PREDICT, PDP_MS_PREDICT_PRIMAL, PDP_MS_PREDICT_GUARD, PDP_MS_WITH_SOC */
    int log_rows; /* rows per trajectory of the optional iteration log (0 = none) */
    int dtheta_bstride;        /* PDP_MS_PREDICT: dtheta [B][p] (stride p) or shared [p] (stride 0) ... */
    const double* dtheta;
    const double* dxdp;        /* ... [B][T+1][n][p] */
    const double* dudp;        /* ... [B][T][m][p] */
    const double* riccati;     /* ... [B][T][pdp_oc_riccati_doubles()] or NULL */
    const float* predict_record; /* alternatively (takes precedence; dxdp / dudp / riccati are then ignored): the packed fp32 prediction record
                                    [B][T][pdp_oc_predict_record_floats()] of pdp_oc_pdp_grad_sens_batched - 74 MB at C3 where the fp64 outputs are 181 MB */
} pdp_oc_ms_opts;
int64_t pdp_oc_solve_ms_workspace_bytes(int B, int T, int max_iter);
/* iter_log (optional, [B][opts->log_rows][8]): one row per accepted step, the columns of IPOPT's iteration output (print_level 5):
 * iteration, objective, inf_pr, inf_du, dw (Hessian shift), alpha, grad(phi)'d, theta = |c|_1. */
int pdp_oc_solve_ms_batched(int B, int T, const double* x0, const double* theta, int theta_bstride, double* x, double* u, double* lam,
                            double* cost, double* resid, int32_t* converged, int32_t* iterations, int32_t* status, double* gains,
                            double* iter_log, const pdp_oc_ms_opts* opts, void* workspace, int64_t workspace_bytes, void* stream);

/* Fused "forward + Riccati + PDP gradient" unit (the loop body of the IRL drivers,
 * Examples/IRL/cartpole/cartpole_PDP.py:45-74, with ocSolver replaced by the given controls, or by the
 * caller's optimal (x,u,lam) when flags has PDP_OC_GIVEN_TRAJ):
 *   rollout -> costates -> aux system in LDS (never in HBM) -> lqrSolver backward/forward on MFMA tiles ->
 *   loss = |x-x_demo|^2 + |u-u_demo|^2 ; grad = sum_t (x_t-xd_t)^T X_t + (u_t-ud_t)^T U_t + (x_T-xd_T)^T X_T
 * Inputs : x0 [B][n], u [B][T][m], theta, demo_x [B][T+1][n], demo_u [B][T][m]
 * In/out : x [B][T+1][n], lam [B][T][n]  (outputs, or inputs with PDP_OC_GIVEN_TRAJ; never NULL)
 * Outputs: loss [B], grad [B][p], optional dxdp [B][T+1][n][p], dudp [B][T][m][p] (NULL = not stored),
 *          status [B].  workspace: pdp_oc_pdp_workspace_bytes(B,T). */
#define PDP_OC_GIVEN_TRAJ 1
#define PDP_OC_PACKED 2 /* grad is [B][p + 1]: gradient and, in the last column, the loss (the row the multi-GPU iteration all-gathers) */
#define PDP_OC_RECORD_PRIMAL 4 /* pdp_oc_pdp_grad_sens_batched: only the X | U part of sens->predict_record is written (same stride; the P | W part is left untouched) -
                                  the record of a prediction of states and controls only (PDP_MS_PREDICT_PRIMAL), 5 instead of 13 stores per stage */
int64_t pdp_oc_pdp_workspace_bytes(int B, int T);
int pdp_oc_pdp_grad_batched(int B, int T, int flags, const double* x0, const double* u, const double* theta,
                            int theta_bstride, const double* demo_x, const double* demo_u, double* x, double* lam,
                            double* loss, double* grad, double* dxdp, double* dudp, int32_t* status,
                            void* workspace, int64_t workspace_bytes, void* stream);

/* The same unit with the solution of the auxiliary control system kept: besides dxdp / dudp (X_t, U_t = state_traj_opt / control_traj_opt of
 * LQR.lqrSolver, PDP.py:611-613) the Riccati matrices of every stage, riccati [B][T][n n + n p + 1] = P_{t+1} [n][n] | W_{t+1} [n][p] | one scratch word
 * (PP[t], WW[t] of PDP.py:561-580; the costate sensitivities are Lambda_t = P_{t+1} X_{t+1} + W_{t+1}, PDP.py:604).  Any of the three may be NULL.
 * pdp_oc_riccati_doubles() = n n + n p + 1 of this model. */
int64_t pdp_oc_riccati_doubles(void);
/* Optional outputs of pdp_oc_pdp_grad_sens_batched; any pointer may be NULL.  predict_record [B][T][pdp_oc_predict_record_floats()]: what the prediction of the
 * next starting point needs, packed and in single precision per stage - X_{t+1} [n][p] | U_t [m][p] | upper triangle of P_{t+1} (row-major packed) | W_{t+1} [n][p]
 * (a starting point is first-order accurate in dtheta at best: fp32 factors are far below that error, and the record stays in the Infinity Cache where the fp64
 * outputs are read at HBM speed). */
typedef struct pdp_oc_sens_out {
    double* dxdp;
    double* dudp;
    double* riccati;
    float* predict_record;
} pdp_oc_sens_out;
int64_t pdp_oc_predict_record_floats(void);
int pdp_oc_pdp_grad_sens_batched(int B, int T, int flags, const double* x0, const double* u, const double* theta,
                                 int theta_bstride, const double* demo_x, const double* demo_u, double* x, double* lam,
                                 double* loss, double* grad, const pdp_oc_sens_out* sens, int32_t* status,
                                 void* workspace, int64_t workspace_bytes, void* stream);

/* First-order prediction of the optimal trajectory at theta + dtheta from those outputs - the derivative the auxiliary control system IS
 * (PDP.py:582-608), used as the starting point of the next OCSys.ocSolver call of an IRL loop (Examples/IRL/quadrotor/uav_PDP.py:52-62: theta moves by
 * lr * gradient per iteration) so that pdp_oc_solve_ms_batched with PDP_MS_WARM needs one Newton iteration fewer:
 *     x_t += X_t dtheta,  u_t += U_t dtheta,  lam_t += P_{t+1} (X_{t+1} dtheta) + W_{t+1} dtheta          in place
 * dtheta [B][p] (dtheta_bstride = p) or shared [p] (stride 0).  riccati and lam may both be NULL: states and controls only. */
int pdp_oc_predict_batched(int B, int T, const double* dtheta, int dtheta_bstride, const double* dxdp, const double* dudp,
                           const double* riccati, double* x, double* u, double* lam, void* stream);
/* the same from the packed fp32 record (lam may be NULL: states and controls only) */
int pdp_oc_predict_record_batched(int B, int T, const double* dtheta, int dtheta_bstride, const float* predict_record, double* x, double* u,
                                  double* lam, void* stream);

/* ---- PDP_KIND_CP ------------------------------------------------------------------------------------ */

/* Policy descriptor: Lagrange polynomial (ControlPlanning.setPolyControl, PDP.py:699-725; theta =
 * vcat(U_0..U_N), p = (N+1) m) or tanh MLP (setNeuralPolicy, 727-759; theta = [vec_F(A_0), b_0, vec_F(A_1), b_1, ...],
 * column-major vec, layers = hidden + [m]). */
#define PDP_POLICY_POLY 0
#define PDP_POLICY_MLP 1
#define PDP_POLICY_TABLE 2   /* u_t = sum_i table[t][i] theta[i m .. i m + m): an open-loop policy given by its basis values at every step.  recmat_* (PDP.py:1081-1141):
                                table[t][i] = 1 where step t lies in grid cell i (theta = the control of every cell); warp_* (PDP.py:960-1008): the Lagrange basis on the cell
                                index.  Served by pdp_cp_integrate_batched and pdp_cp_step_batched (size-generic adjoint kernel) */
typedef struct pdp_policy {
    int kind;
    int n_pivots;            /* POLY: number of pivots (<= 16) */
    double pivots[16];       /* POLY: pivot times */
    int n_layers;            /* MLP: number of weight layers (<= 16), sizes[k] = rows of A_k, last = m.  Up to 4 layers of <= 16 units: register-resident kernel; up to 8 of
                                <= 32 (p <= 512): LDS-resident adjoint kernel; anything else: the size-generic kernel (csrc/pdp_cp_generic_kernels.h) */
    int sizes[16];
    int n_basis;             /* TABLE: number of basis functions (p = n_basis * m) */
    const double* table;     /* TABLE: device memory, [T][n_basis] row-major */
} pdp_policy;

/* ControlPlanning.integrateSys (PDP.py:763-786): x0 [B][n], theta -> x [B][T+1][n], u [B][T][m], cost [B]. */
int pdp_cp_integrate_batched(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* theta,
                             int theta_bstride, double* x, double* u, double* cost, void* stream);

/* ControlPlanning.getAuxSys (PDP.py:788-811): dynF [B][T][n][n], dynG [B][T][n][m], dUx [B][T][m][n], dUe [B][T][m][p];
 * plus, optionally (NULL = skipped), the cost gradients that ControlPlanning.step evaluates along the trajectory
 * (dcx_fn / dcu_fn / dhx_fn, PDP.py:871-876): dcx [B][T][n], dcu [B][T][m], dhx [B][n].  x is [B][T+1][n]. */
int pdp_cp_auxsys_batched(int B, int T, const pdp_policy* pol, int p, const double* x, const double* u, const double* theta,
                          int theta_bstride, double* dynF, double* dynG, double* dUx, double* dUe, double* dcx, double* dcu,
                          double* dhx, void* stream);

/* ControlPlanning.step (PDP.py:850-878), fused: loss [B] = sum c + h, grad [B][p] = sum_t c_x X_t + c_u U_t + h_x X_T;
 * optional x [B][T+1][n], u [B][T][m] (NULL = not stored).  PDP_POLICY_POLY with p <= 64: forward sensitivities on MFMA tiles
 * (the reference's own formulation); PDP_POLICY_MLP (<= 8 layers of <= 32 units, p <= 512) and larger Lagrange policies: the same
 * gradient by one adjoint sweep, O(T (n^2 + p)) instead of O(T n^2 p).  The materialised route of the reference
 * (integrate -> auxsys -> pdp_cp_aux_integrate_batched -> pdp_cp_grad_contract_batched) stays available for getAuxSys/integrateAuxSys.
 * workspace (optional, NULL = none): pdp_cp_step_workspace_bytes(B,T,pol,p) bytes; the hidden activations of an MLP policy live there
 * (register-resident kernel: 64 doubles per time step; general kernel: when LDS residency would limit occupancy; 0 bytes = not needed).
 * Without it an MLP policy runs on the general kernel. */
int64_t pdp_cp_step_workspace_bytes(int B, int T, const pdp_policy* pol, int p);
int pdp_cp_step_batched(int B, int T, const pdp_policy* pol, int p, const double* x0, const double* theta, int theta_bstride,
                        double* loss, double* grad, double* x, double* u, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- PDP_KIND_SYSID --------------------------------------------------------------------------------- */

/* SysID.integrateDyn (PDP.py:1209-1223): x0 [B][n], u [B][T][m] -> x [B][T+1][n]. */
int pdp_sysid_integrate_batched(int B, int T, const double* x0, const double* u, const double* theta, int theta_bstride,
                                double* x, void* stream);

/* SysID.getAuxSys (PDP.py:1225-1239): dynF [B][T][n][n], dynE [B][T][n][p]. */
int pdp_sysid_auxsys_batched(int B, int T, const double* x, const double* u, const double* theta, int theta_bstride,
                             double* dynF, double* dynE, void* stream);

/* SysID.step per trajectory (PDP.py:1261-1296 without the final mean): u [B][T][m], x_obs [B][T+1][n] ->
 * loss [B] = |x - x_obs|^2, grad [B][p] = sum_{t<=T} (x_t - x_obs_t)^T X_t  (= half the gradient, as in the
 * reference); the caller averages over the batch (PDP.py:1293-1294). */
int pdp_sysid_step_batched(int B, int T, const double* u, const double* x_obs, const double* theta, int theta_bstride,
                           double* loss, double* grad, void* stream);
/* The same with a caller-owned workspace of pdp_sysid_step_workspace_bytes(B, T) bytes (0 for batches that do not use one): batches with more than two trajectories per
 * SIMD roll their trajectories out beforehand with one LANE per trajectory (the pass behind pdp_sysid_integrate_batched, into the workspace [B][T+1][n]) and run the fused
 * kernel on them - inside the fused kernel a rollout occupies a whole wavefront per trajectory.  Same results. */
int64_t pdp_sysid_step_workspace_bytes(int B, int T);
int pdp_sysid_step_ws_batched(int B, int T, const double* u, const double* x_obs, const double* theta, int theta_bstride,
                              double* loss, double* grad, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDP_HIP_H */
