"""Batched optimal-control solve on the GPU - stands where the reference's OCSys.ocSolver hands a multiple-shooting NLP to
IPOPT (PDP/PDP.py:121-220).

Default path (solve_batch with no starting controls): the reference's own formulation - the multiple-shooting NLP from the
all-zero initial guess (PDP.py:155,166), iterated the way IPOPT does (primal-dual Newton step, inertia correction, filter line
search; csrc/pdp_ocsolve2_kernels.h, pdp_oc_solve_ms_batched): a persistent pair of wavefronts per trajectory, every iteration inside
one launch.  It reproduces the optima the reference stored from a cold start on all five benchmark systems.  A trajectory whose
line search falls below IPOPT's alpha_min is restored inside the kernel (states <- rollout of its controls; include/pdp_hip.h); whatever still
comes back unconverged (no restoration possible, iteration limit) falls back to the single-shooting solver below.

Single-shooting path (starting controls given, or as the fallback): stagewise Newton / iLQR, every trajectory of the batch in
parallel; the iteration loop runs inside the model library (pdp_oc_solve_batched, see _solve below), plus the choice of the
starting controls and a batch-level globalisation (solve_batch_single_shooting):

    repeat:  costates  lambda = c_x + f_x' lambda+ , stationarity residual H_u
             F, G, Hxx, Hxu, Huu, hxx along (x, u, lambda)
             LQ sub-problem for (dx, du): the same Riccati kernel as LQR.lqrSolver with p = 1, Hue := H_u
             closed-loop rollouts u = ubar - alpha k - K (x - xbar), backtracking on the true cost

Hessians: the full Hamiltonian Hessians give Newton's method (quadratic convergence, what makes the multipliers accurate to
1e-10); where that step fails (indefinite Quu, no decrease) the sample falls back to the Gauss-Newton (iLQR) Hessians
- the same kernels evaluated with lambda = 0 - plus Levenberg-Marquardt damping.  At convergence H_u = 0 and
(x, u, lambda) satisfy the same KKT conditions IPOPT solves, so `costate[t] = lambda_{t+1}` matches `lam_g`.
"""
import numpy as np

from . import runtime

# stationarity residual (relative to the control magnitude) below which a sample switches from Gauss-Newton to Newton steps
# (swept on 256-problem batches of cart-pole / robot arm / quadrotor: switching earlier stalls the swing-up problems)
NEWTON_SWITCH = 1e-2


def _solve(oc, ini_state, horizon, auxvar_value, u_init=None, tol=1e-9, max_iter=300, print_level=0, force_init=False, want_gains=False,
           straggler_patience=0):
    """oc: PDP.OCSys.  ini_state [B,n]; auxvar_value [p] or [B,p]; returns dict of CUDA tensors
    state [B,T+1,n], control [B,T,m], costate [B,T,n], cost [B], grad_norm [B], iterations (int), converged [B] (bool).
    u_init (optional warm start) is used per sample only where its rollout is finite and cheaper than u = 0.
    The iterations run inside the model library (pdp_oc_solve_batched, csrc/pdp_model.hip: oc_solve), which sequences

        costates  lambda = c_x + f_x' lambda+ , stationarity residual H_u                  (oc_costate / oc_auxsys kernels)
        per sample: converged?  Newton or Gauss-Newton (Hessians at lambda = 0)?           (oc_newton_prepare_kernel)
        F, G, Hxx, Hxu, Huu + mu I, hxx                                                    (oc_auxsys kernel)
        LQ sub-problem for (dx, du): the LQR.lqrSolver kernel with p = 1, Hue := H_u       (lqr_solve_kernel)
        closed-loop rollouts u = ubar - alpha k - K (x - xbar) for alpha = 1, 1/2, ...     (oc_linesearch_kernel)
        Armijo selection, step acceptance, damping / mode update                           (oc_ls_select_kernel)

    on the stream without a host round trip per iteration; Python only picks the starting controls."""
    torch = runtime.torch_cuda()
    mdl = oc.model()
    n, m = mdl.n, mdl.m
    x0 = runtime.dev(ini_state).reshape(-1, n)
    B, T = x0.shape[0], int(horizon)
    th = oc._theta(auxvar_value, B)
    u = torch.zeros((B, T, m), dtype=torch.float64, device="cuda")
    if u_init is not None:
        uw = runtime.dev(u_init).reshape(B, T, m)
        if force_init:
            xw, Jw = mdl.oc_rollout(x0, uw, th)
            better = torch.isfinite(Jw) & torch.isfinite(xw).all(dim=(1, 2))
        else:
            _, J = mdl.oc_rollout(x0, u, th)
            xw, Jw = mdl.oc_rollout(x0, uw, th)
            better = torch.isfinite(Jw) & torch.isfinite(xw).all(dim=(1, 2)) & ((Jw < J) | ~torch.isfinite(J))
        u = torch.where(better.view(B, 1, 1), uw, u)
    return mdl.oc_solve(x0, u, th, tol=tol, newton_switch=NEWTON_SWITCH, max_iter=max_iter, straggler_patience=straggler_patience,
                        print_level=print_level, want_gains=want_gains)


def solve_batch(oc, ini_state, horizon, auxvar_value, u_init=None, tol=1e-9, max_iter=300, print_level=0, neighbor_retries=2,
                warm_start=None, want_gains=False, method="auto", predict=None, watchdog=False):
    """watchdog=True: the multiple-shooting kernel runs with IPOPT's watchdog (PDP_MS_WITH_WATCHDOG: opt-in, see include/pdp_hip.h) - fewer cold solves of the crawling
    kind are left to the single-shooting fallback (rocket, T = 100: 97 instead of 129 of 512).  Ignored on the other routes.
    ocSolver for a batch.  method "auto": the multiple-shooting solver (the reference's formulation, IPOPT's iteration) unless
    starting controls `u_init` are given; "ms" / "single" force one ("ms" with `u_init`: the multiple-shooting iteration started from those controls, their rollout
    and the least-squares multipliers).  warm_start: a previous solution of the same batch (dict with
    state, control, costate[, gains]) - the multiple-shooting solver starts from that point.  Samples the multiple-shooting solver
    returns unconverged (no restoration possible, iteration limit) are re-solved by the single-shooting path.
    predict (with warm_start = the solution at the PREVIOUS parameter): dict(dtheta, record) - the parameter step and the packed prediction record of
    OCSys.pdp_grad_batch(..., want_predict_record=True) at that solution - or dict(dtheta, dxdp, dudp[, riccati]) with its fp64 sensitivity outputs; the solver then starts from the first-order prediction
    (x, u, lam) + (X, U, Lambda) dtheta (PDP_MS_PREDICT) and saves a Newton iteration - the IRL loop of examples/irl_pdp.py.
    Returns dict of CUDA tensors: state [B,T+1,n], control [B,T,m], costate [B,T,n], cost [B], grad_norm [B], converged [B] (bool),
    iterations (int, sequential iterations of the slowest sample), method_ms [B] (bool: solved by the multiple-shooting kernel)."""
    torch = runtime.torch_cuda()
    mdl = oc.model()
    big = mdl.n > 16 or mdl.m > 4                         # beyond the multiple-shooting kernel's tiles
    generic = big and method != "single" and u_init is None and warm_start is None
    if not generic and (method == "single" or big or (method == "auto" and u_init is not None) or (warm_start is not None and "costate" not in warm_start)):
        if predict is not None:
            import warnings
            warnings.warn("ocsolver.solve_batch: `predict` is used by the multiple-shooting kernel only; this call takes single shooting (method / u_init / a warm start "
                          "without multipliers) and starts from the warm start as it is", RuntimeWarning)
        return solve_batch_single_shooting(oc, ini_state, horizon, auxvar_value, u_init=u_init, tol=tol, max_iter=max_iter, print_level=print_level,
                                           neighbor_retries=neighbor_retries, warm_start=warm_start, want_gains=want_gains)
    x0 = runtime.dev(ini_state).reshape(-1, mdl.n)
    B = x0.shape[0]
    th = oc._theta(auxvar_value, B)
    if generic:
        # the same NLP and iteration, kernel by kernel (solve_batch_ms_generic)
        if predict is not None:
            import warnings
            warnings.warn("ocsolver.solve_batch: `predict` is not used on the kernel-by-kernel route (models beyond n = 16 / m = 4)", RuntimeWarning)
        ms = solve_batch_ms_generic(oc, ini_state, horizon, auxvar_value, tol=min(tol, 1e-9) * 0.1, max_iter=max_iter, print_level=print_level)
    else:
        warm = None if warm_start is None else (warm_start["state"], warm_start["control"], warm_start["costate"])
        ms = mdl.oc_solve_ms(x0, th, horizon, tol=min(tol, 1e-9) * 0.1, max_iter=max_iter, warm=warm, want_gains=want_gains,
                             u_init=u_init if warm is None else None,      # method "ms" with starting controls: PDP_MS_FROM_CONTROLS
                             predict=predict if warm is not None else None, watchdog=watchdog)
    sol = {"state": ms["state"], "control": ms["control"], "costate": ms["costate"], "cost": ms["cost"], "grad_norm": ms["resid"][:, 1].contiguous(),
           "converged": ms["converged"], "iterations": int(ms["iterations"].max()), "method_ms": ms["converged"].clone(), "status": ms["status"]}
    if want_gains:
        if generic:
            # (the LQR gains of the last Newton step are not kept on the kernel-by-kernel route: one single-shooting refinement step at the solution provides them)
            sub = solve_batch_single_shooting(oc, ini_state, horizon, auxvar_value, u_init=ms["control"], tol=tol, max_iter=max_iter, print_level=print_level,
                                              neighbor_retries=0, want_gains=True)
            sol["gains"] = sub["gains"]
        else:
            sol["gains"] = ms["gains"]
    bad = ~ms["converged"]
    if print_level > 0:
        print("  multiple-shooting solve: %d/%d converged, max %d iterations" % (int(ms["converged"].sum()), B, sol["iterations"]))
    if bool(bad.any()) and not (generic and method == "ms"):
        # ONLY the rows the multiple-shooting iteration left unconverged (no restoration possible, iteration limit) go on to single shooting; rows
        # that converged keep IPOPT's optimum (a single-shooting solve from u = 0 may end in another basin - the rocket)
        bi = torch.nonzero(bad).flatten()
        th_np = np.asarray(th, dtype=np.float64).reshape(-1, oc.n_auxvar)
        thb = th_np[bi.cpu().numpy()] if th_np.shape[0] == B and B > 1 else th_np[0]
        u0 = warm_start["control"][bi] if warm_start is not None else None
        sub = solve_batch_single_shooting(oc, x0[bi], horizon, thb, u_init=u0, tol=tol, max_iter=max_iter, print_level=print_level,
                                          neighbor_retries=neighbor_retries, want_gains=want_gains)
        # a row is replaced only where the fallback did better: it converged, or - both unconverged - it ended at a finite point with a
        # smaller stationarity residual than the multiple-shooting iterate (a non-finite or worse fallback result never overwrites it)
        fin_sub = torch.isfinite(sub["cost"]) & torch.isfinite(sub["grad_norm"]) & torch.isfinite(sub["state"]).all(dim=(1, 2))
        fin_ms = torch.isfinite(sol["cost"][bi]) & torch.isfinite(sol["grad_norm"][bi])
        take = sub["converged"] | (fin_sub & (~fin_ms | (sub["grad_norm"] < sol["grad_norm"][bi])))
        idx = bi[take]
        for k in ("state", "control", "costate", "cost", "grad_norm", "converged") + (("gains",) if want_gains else ()):
            sol[k][idx] = sub[k][take]
        sol["status"][idx] = 0                                 # the multiple-shooting status bits no longer describe these rows
        sol["iterations"] += int(sub["iterations"])
    return sol


def solve_batch_ms_generic(oc, ini_state, horizon, auxvar_value, tol=1e-10, max_iter=300, print_level=0, restoration=True, log_rows=0, soc=False):
    """The multiple-shooting NLP of OCSys.ocSolver (PDP.py:131-182) iterated the way IPOPT does - the algorithm of pdp_oc_solve_ms_batched and of its CPU restatement
    oracle/ipopt_ms.py - for problems BEYOND that kernel's tiles (16 < n <= 32 or 4 < m <= 8), kernel by kernel: residuals of the NLP (pdp_oc_ms_residuals_batched),
    KKT matrices (pdp_oc_auxsys_batched), the Newton step as an LQ problem with one affine column on the generic LQR kernel (pdp_lqr_solve_batched, whose status
    reports whether every Quu was positive definite = the inertia test), the restoration by rollout (pdp_oc_rollout_batched).  Only IPOPT's bookkeeping - inertia
    correction schedule, filter, step lengths, per sample - is tensor arithmetic here.  Several launches per iteration: a route for sizes the fast kernel does not
    take (before round 3 these problems were solved by single shooting, i.e. a different iterate path), not a fast path.
    soc=True: IPOPT's second-order correction (include/pdp_hip.h, off by default as in the kernel; oracle/ipopt_ms.py) - per sample, up to four corrected steps when the first trial point is rejected
    without progress towards feasibility; each is one more solve of the LQ problem with the accumulated constraint block.
    Returns the dict of ModelLib.oc_solve_ms (status bits PDP_MS_*; log [B, log_rows, 8] with log_rows > 0)."""
    torch = runtime.torch_cuda()
    mdl = oc.model()
    n, m, T = mdl.n, mdl.m, int(horizon)
    x0 = runtime.dev(ini_state).reshape(-1, n)
    B = x0.shape[0]
    th = oc._theta(auxvar_value, B)
    f64 = dict(dtype=torch.float64, device="cuda")
    x, u, lam = torch.zeros((B, T + 1, n), **f64), torch.zeros((B, T, m), **f64), torch.zeros((B, T, n), **f64)       # w0 = 0 (PDP.py:155,166) ...
    x[:, 0] = x0                                                                                                      # ... x_0 = ini_state
    eye_n, eye_m = torch.eye(n, **f64), torch.eye(m, **f64)
    G_TH, G_PH, EPS = 1e-5, 1e-8, 2.220446049250313e-16                                                               # gamma_theta, gamma_phi (IPOPT defaults)

    def resid(x_, u_, l_):
        r = mdl.oc_ms_residuals(x_, u_, l_, th)
        r["f"], r["theta"], r["inf_pr"] = r["cost"].sum(1), r["c"].abs().sum((1, 2)), r["c"].abs().amax((1, 2))
        r["inf_du"] = torch.maximum(r["rx"].abs().amax((1, 2)), r["ru"].abs().amax((1, 2)))
        return r

    def matrices(x_, u_, l_):
        return mdl.oc_auxsys(x_, u_, l_, th, only=("dynF", "dynG", "Hxx", "Hxu", "Huu", "hxx"))

    def kkt(A, r, dw, ls=False):
        """Newton step of the KKT system [W + dw I, A'; A, 0] as the LQ problem with affine terms; ls: the least-squares multiplier estimate (W = I, no defects)."""
        if ls:
            Hxx, Huu, Hxu, hxx, E = eye_n.expand(B, T, n, n).contiguous(), eye_m.expand(B, T, m, m).contiguous(), None, eye_n.expand(B, n, n).contiguous(), None
        else:
            Hxx, Huu, Hxu = A["Hxx"] + dw.view(B, 1, 1, 1) * eye_n, A["Huu"] + dw.view(B, 1, 1, 1) * eye_m, A["Hxu"]
            hxx, E = A["hxx"] + dw.view(B, 1, 1) * eye_n, r["c"].unsqueeze(-1).contiguous()
        X, U, L, st = runtime.lqr_solve(A["dynF"], A["dynG"], Hxx, Huu, hxx, r["rx"][:, T].unsqueeze(-1).contiguous(), E=E, Hxu=Hxu,
                                        Hxe=r["rx"][:, :T].unsqueeze(-1).contiguous(), Hue=r["ru"].unsqueeze(-1).contiguous(), T=T)
        return X[..., 0], U[..., 0], L[..., 0], (st & (256 | 2 | 1)) == 0

    def put(dst, src, mask):
        dst.copy_(torch.where(mask.view(-1, *([1] * (dst.dim() - 1))), src, dst))

    def ls_multipliers(x_, u_, r):
        _, _, l0, _ = kkt(matrices(x_, u_, torch.zeros_like(lam)), r, None, ls=True)
        good = torch.isfinite(l0).all(dim=(1, 2)) & (l0.abs().amax((1, 2)) <= 1000.0)                                # constr_mult_init_max
        return torch.where(good.view(B, 1, 1), l0, torch.zeros_like(l0))

    r = resid(x, u, lam)
    lam = ls_multipliers(x, u, r)
    r = resid(x, u, lam)
    theta_max, theta_min = 1e4 * r["theta"].clamp(min=1.0), 1e-4 * r["theta"].clamp(min=1.0)
    K = max_iter + 2
    fth, fph, nf = torch.zeros((B, K), **f64), torch.zeros((B, K), **f64), torch.zeros((B,), dtype=torch.int64, device="cuda")
    kidx = torch.arange(K, device="cuda").view(1, K)
    conv, status = torch.zeros((B,), dtype=torch.bool, device="cuda"), torch.zeros((B,), dtype=torch.int32, device="cuda")
    restored, corrected = torch.zeros_like(conv), torch.zeros_like(conv)
    iters, dw_last = torch.zeros((B,), dtype=torch.int32, device="cuda"), torch.zeros((B,), **f64)
    log = torch.zeros((B, int(log_rows), 8), **f64) if log_rows > 0 else None
    rkeys = ("c", "rx", "ru", "cost", "f", "theta", "inf_pr", "inf_du")

    def filter_add(mask, theta, f):
        rows = torch.nonzero(mask).flatten()
        fth[rows, nf[rows]] = (1.0 - G_TH) * theta[rows]
        fph[rows, nf[rows]] = f[rows] - G_PH * theta[rows]
        nf[rows] += 1

    for it in range(max_iter + 1):
        active = ~conv & (status == 0)
        scale = 1.0 + torch.maximum(x.abs().amax((1, 2)), u.abs().amax((1, 2)))
        lscale = 1.0 + lam.abs().amax((1, 2))
        fin = torch.isfinite(r["f"]) & torch.isfinite(r["inf_pr"]) & torch.isfinite(r["inf_du"])
        status = torch.where(active & ~fin, status | 1, status)                                                      # PDP_STATUS_NONFINITE
        conv = conv | (active & fin & (r["inf_pr"] <= tol * scale) & (r["inf_du"] <= tol * lscale))
        active = ~conv & (status == 0)
        if not bool(active.any()):
            break
        if it == max_iter:
            status = torch.where(active, status | 8, status)                                                         # PDP_MS_MAXITER
            break
        iters = torch.where(active, iters + 1, iters)
        # ---- search direction with inertia correction (Algorithm IC), every sample on its own dw
        dw, need = torch.zeros((B,), **f64), active.clone()
        dx, du, dl = torch.zeros_like(x), torch.zeros_like(u), torch.zeros_like(lam)
        A = matrices(x, u, lam)
        while True:
            X, U, L, ok = kkt(A, r, dw)
            got = need & ok
            put(dx, X, got); put(du, U, got); put(dl, L, got)
            need = need & ~ok
            if not bool(need.any()):
                break
            first = dw_last == 0.0
            dw_new = torch.where(dw == 0.0, torch.where(first, torch.full_like(dw, 1e-4), (dw_last / 3.0).clamp(min=1e-20)), dw * torch.where(first, 100.0, 8.0))
            dw = torch.where(need, dw_new, dw)
            dead = need & (dw > 1e20)
            status = torch.where(dead, status | 16, status)                                                          # PDP_MS_INERTIA
            need = need & ~dead
            if not bool(need.any()):
                break
        active = active & (status == 0)
        dw_last = torch.where(active & (dw > 0.0), dw, dw_last)
        gd = (r["rx"] * dx).sum((1, 2)) + (r["ru"] * du).sum((1, 2)) + (lam * r["c"]).sum((1, 2))                     # grad phi' d = rd' d + lam' c   (A d = -c)
        # ---- backtracking filter line search (Algorithm A)
        f, theta = r["f"], r["theta"]
        neg = gd < 0.0
        ngd = (-gd).clamp(min=1e-300)
        amin = torch.where(neg, torch.minimum(torch.full_like(gd, G_TH), G_PH * theta / ngd), torch.full_like(gd, G_TH))
        amin = torch.where(neg & (theta <= theta_min), torch.minimum(amin, theta ** 1.1 / ngd ** 2.3), amin) * 0.05
        alpha, searching = torch.ones((B,), **f64), active.clone()
        accepted, ftype, soc_taken = torch.zeros_like(conv), torch.zeros_like(conv), torch.zeros_like(conv)
        xn, un, ln = x.clone(), u.clone(), lam.clone()
        rn = {k: r[k].clone() for k in rkeys}

        def acceptable(rt, who):
            """trial points rt of the samples `who`, tested with the step length alpha (steps A-5.3 / A-5.4): (accepted, f-type)"""
            ft, tht = rt["f"], rt["theta"]
            dominated = ((kidx < nf.view(B, 1)) & (tht.view(B, 1) >= fth) & (ft.view(B, 1) >= fph)).any(dim=1)
            okf = torch.isfinite(ft) & torch.isfinite(tht) & (tht <= theta_max) & ~dominated
            switching = neg & (alpha * ngd ** 2.3 > theta ** 1.1)
            c1 = (theta <= theta_min) & switching
            acc = who & okf & torch.where(c1, ft <= f + 1e-8 * alpha * gd + 10.0 * EPS * f.abs(), (tht <= (1.0 - G_TH) * theta) | (ft <= f - G_PH * theta))
            return acc, acc & c1

        def take(xt, ut, lt, rt, acc):
            put(xn, xt, acc); put(un, ut, acc); put(ln, lt, acc)
            for k in rkeys:
                put(rn[k], rt[k], acc)

        first_trial = True
        while bool(searching.any()):
            a3 = alpha.view(B, 1, 1)
            xt, ut, lt = x + a3 * dx, u + a3 * du, lam + a3 * dl
            rt = resid(xt, ut, lt)
            acc, ft_ = acceptable(rt, searching)
            take(xt, ut, lt, rt, acc)
            ftype, accepted = ftype | ft_, accepted | acc
            searching = searching & ~acc
            if first_trial and soc:
                # second-order correction: same matrix (same dw), constraint block c_soc = alpha c(x_k) + c(x_k + alpha d), accumulated; alpha = 1 here
                live = searching & torch.isfinite(rt["f"]) & torch.isfinite(rt["theta"]) & (rt["theta"] >= theta)
                c_soc, ct, th_old = r["c"].clone(), rt["c"], rt["theta"].clone()
                for _ in range(4):
                    if not bool(live.any()):
                        break
                    c_soc = torch.where(live.view(B, 1, 1), c_soc + ct, c_soc)
                    rs = dict(r)
                    rs["c"] = c_soc
                    X, U, L, ok = kkt(A, rs, dw)
                    live = live & ok & torch.isfinite(X).all(dim=(1, 2)) & torch.isfinite(U).all(dim=(1, 2)) & torch.isfinite(L).all(dim=(1, 2))
                    xt2, ut2, lt2 = x + X, u + U, lam + L
                    rt2 = resid(xt2, ut2, lt2)
                    acc2, ft2_ = acceptable(rt2, live)
                    take(xt2, ut2, lt2, rt2, acc2)
                    ftype, accepted, soc_taken = ftype | ft2_, accepted | acc2, soc_taken | acc2
                    searching = searching & ~acc2
                    live = live & ~acc2 & torch.isfinite(rt2["f"]) & torch.isfinite(rt2["theta"]) & (rt2["theta"] <= 0.99 * th_old)
                    th_old, ct = torch.where(live, rt2["theta"], th_old), rt2["c"]
            first_trial = False
            alpha = torch.where(searching, alpha * 0.5, alpha)
            searching = searching & (alpha >= amin) & (alpha > 1e-300)
        if log is not None and it < log_rows:
            row = torch.stack([torch.full_like(f, float(it)), f, r["inf_pr"], r["inf_du"], dw,
                               torch.where(accepted, torch.where(soc_taken, -alpha, alpha), torch.zeros_like(alpha)), gd, theta], dim=1)
            log[:, it] = torch.where(active.view(B, 1), row, log[:, it])
        failed = active & ~accepted
        corrected = corrected | soc_taken
        filter_add(accepted & ~ftype, theta, f)
        x, u, lam, r = xn, un, ln, rn
        if bool(failed.any()):
            # ---- restoration (oracle/ipopt_ms.py: solve; include/pdp_hip.h): current point into the filter, states <- rollout of the controls, multipliers reset
            can = failed & (theta > 0.0) if restoration else torch.zeros_like(failed)
            status = torch.where(failed & ~can, status | 4, status)                                                   # PDP_MS_RESTORATION
            if bool(can.any()):
                filter_add(can, theta, f)
                xr, _ = mdl.oc_rollout(x0, u, th, want_cost=False)
                finr = torch.isfinite(xr).all(dim=(1, 2))
                status = torch.where(can & ~finr, status | 4, status)
                can = can & finr
                put(x, xr, can)
                put(lam, torch.zeros_like(lam), can)
                r0 = resid(x, u, lam)
                l0 = ls_multipliers(x, u, r0)
                put(lam, l0, can)
                r1 = resid(x, u, lam)
                for k in rkeys:
                    put(r[k], r1[k], can)
                restored = restored | can
        if print_level > 0:
            print("  ms-generic iteration %3d: %d / %d converged" % (it, int(conv.sum()), B))
    status = torch.where(restored, status | 128, status)                                                             # PDP_MS_RESTORED (informational)
    status = torch.where(corrected, status | 1024, status)                                                           # PDP_MS_SOC (informational)
    out = {"state": x, "control": u, "costate": lam, "cost": r["f"], "resid": torch.stack([r["inf_pr"], r["inf_du"]], dim=1), "converged": conv,
           "iterations": iters, "status": status}
    if log is not None:
        out["log"] = log
    return out


def solve_batch_single_shooting(oc, ini_state, horizon, auxvar_value, u_init=None, tol=1e-9, max_iter=300, print_level=0, neighbor_retries=2,
                                warm_start=None, want_gains=False):
    """Batched single-shooting OC solve (see _solve) plus a batch-level globalisation: samples that did not converge (non-convex problems such
    as the cart-pole swing-up can trap single shooting in a poor basin) are re-solved from a CLOSED-LOOP warm start: the optimal
    trajectory and LQR feedback gains of the nearest converged sample (distance in initial state and parameter) are rolled out
    from the stuck sample's own initial state (open-loop warm starts diverge on unstable systems), up to `neighbor_retries` times."""
    torch = runtime.torch_cuda()
    mdl = oc.model()
    force = False
    if warm_start is not None:
        # closed-loop warm start from a previous solution of the SAME batch (dict with state, control, gains - e.g. the previous
        # iterate of an IRL loop, or the solution at a neighbouring parameter): u = ubar - K (x - xbar) rolled out at the new theta
        x0w = runtime.dev(ini_state).reshape(-1, mdl.n)
        _, u_init, _ = mdl.oc_rollout_feedback(x0w, warm_start["control"], warm_start["state"], warm_start["gains"],
                                               torch.zeros((x0w.shape[0],), dtype=torch.float64, device="cuda"), oc._theta(auxvar_value, x0w.shape[0]))
        force = True
    nb = runtime.dev(ini_state).reshape(-1, mdl.n).shape[0]
    sol = _solve(oc, ini_state, horizon, auxvar_value, u_init=u_init, tol=tol, max_iter=max_iter, print_level=print_level, force_init=force,
                 want_gains=want_gains, straggler_patience=6 if (nb > 1 and neighbor_retries > 0) else 0)
    B = sol["state"].shape[0]
    if B == 1 or neighbor_retries <= 0:
        return sol
    x0 = runtime.dev(ini_state).reshape(B, -1)
    th_np = np.asarray(oc._theta(auxvar_value, B), dtype=np.float64).reshape(-1, oc.n_auxvar)
    th = runtime.dev(th_np)
    feat = torch.cat([x0, th.expand(B, -1)], dim=1)
    feat = feat / (feat.abs().amax(dim=0, keepdim=True) + 1e-12)
    for _ in range(neighbor_retries):
        conv = sol["converged"]
        if not bool(conv.any()):
            break
        # candidates for a retry: not converged, or converged into a basin markedly worse than a nearby sample's
        # (cost > 1.25 x the cheapest of the 8 nearest converged neighbours) - single shooting on non-convex problems
        gi = torch.nonzero(conv).flatten()
        dist = torch.cdist(feat, feat[gi])
        k = min(8, gi.numel())
        nn = dist.topk(k, dim=1, largest=False).indices                       # [B,k] indices into gi
        nn_cost = sol["cost"][gi][nn]
        best = nn_cost.argmin(dim=1)
        donor_all = gi[nn.gather(1, best.view(-1, 1)).flatten()]
        worse = conv & (sol["cost"] > 1.25 * nn_cost.min(dim=1).values)
        bad = ~conv | worse
        if not bool(bad.any()):
            break
        bi = torch.nonzero(bad).flatten()
        donors = donor_all[bi]
        thb = th_np[bi.cpu().numpy()] if th_np.shape[0] == B else th_np[0]
        thd = th_np[donors.cpu().numpy()] if th_np.shape[0] == B else th_np[0]
        don = _solve(oc, x0[donors], horizon, thd, u_init=sol["control"][donors], tol=tol, max_iter=3, force_init=True, want_gains=True)
        _, u_cl, _ = mdl.oc_rollout_feedback(x0[bi], don["control"], don["state"], don["gains"], torch.zeros_like(sol["cost"][bi]), thb)
        sub = _solve(oc, x0[bi], horizon, thb, u_init=u_cl, tol=tol, max_iter=max_iter, print_level=print_level, force_init=True)
        better = (sub["converged"] & ~sol["converged"][bi]) | (sub["converged"] & (sub["cost"] < sol["cost"][bi])) | \
                 (~sol["converged"][bi] & (sub["cost"] < sol["cost"][bi]))
        idx = bi[better]
        if want_gains:
            sub2 = _solve(oc, x0[bi], horizon, thb, u_init=sub["control"], tol=tol, max_iter=1, force_init=True, want_gains=True)
            sub["gains"] = sub2["gains"]
        for k in ("state", "control", "costate", "cost", "grad_norm", "converged") + (("gains",) if want_gains else ()):
            sol[k][idx] = sub[k][better]
        sol["iterations"] += don["iterations"] + sub["iterations"] + 2         # sequential iterations executed, retries included
    return sol


BOUND_RELAX = 1e-8            # IPOPT's bound_relax_factor: a finite bound is moved outwards by 1e-8 max(1, |bound|)


def relaxed_bounds(oc, x0=None):
    """(lbx, ubx, lbu, ubu) of the barrier sub-problems.  The caller's bounds, except that a STATE bound some initial state sits on is relaxed the way IPOPT relaxes
    every bound (bound_relax_factor): the reference's NLP does not bound x_0 at all (PDP.py:141-146), but in the barrier sub-problems the term of x_0 is a constant that
    must be finite.  (Relaxing all bounds, as IPOPT does, would let the returned controls exceed theirs by 1e-8 - and projecting them back moves the rollout of an
    unstable system by more than the solver's tolerance; so only where it is needed.)  x0 [B, n] or None (nothing relaxed)."""
    lbx, ubx = np.asarray(oc.state_lb, dtype=float).copy(), np.asarray(oc.state_ub, dtype=float).copy()
    lbu, ubu = np.asarray(oc.control_lb, dtype=float), np.asarray(oc.control_ub, dtype=float)
    if x0 is not None:
        x0 = np.atleast_2d(np.asarray(x0, dtype=float))
        sl, su = BOUND_RELAX * np.maximum(1.0, np.abs(lbx)), BOUND_RELAX * np.maximum(1.0, np.abs(ubx))
        on_l = (np.abs(lbx) < 1e19) & ((x0 >= lbx[None]) & (x0 <= (lbx + sl)[None])).any(axis=0)
        on_u = (np.abs(ubx) < 1e19) & ((x0 <= ubx[None]) & (x0 >= (ubx - su)[None])).any(axis=0)
        lbx, ubx = np.where(on_l, lbx - sl, lbx), np.where(on_u, ubx + su, ubx)
    return lbx, ubx, lbu, ubu


def solve_batch_bounded(oc, ini_state, horizon, auxvar_value, tol=1e-8, max_iter=300, print_level=0, mu0=0.1, **unsupported):
    """ocSolver with finite state / control bounds (the reference hands them to IPOPT as lbw / ubw, PDP/PDP.py:141-168).

    Log-barrier continuation around the equality-constrained multiple-shooting kernel: for mu = 0.1, then mu <- max(mu_min, min(0.2 mu, mu^1.5))
    (IPOPT's monotone barrier schedule, kappa_mu = 0.2, theta_mu = 1.5) the sub-problem  min sum [c + mu b] + h  s.t. the dynamics  - b the log barrier
    of the bounds, part of the generated cost (OCSys.barrier_model: the symbolic front-end differentiates it like any cost term) - is solved by
    pdp_oc_solve_ms_batched, warm-started from the previous mu.  The filter line search keeps the iterates strictly inside the bounds by itself: a
    trial point outside makes the objective non-finite and is rejected (the step is halved).  The first sub-problem starts from IPOPT's starting point:
    the reference's initial guess w0 = (lb + ub) / 2 where both bounds are finite (PDP.py:155,167), 0 otherwise, pushed into the interior (bound_push =
    bound_frac = 1e-2).  A state bound an initial state sits on is relaxed as IPOPT relaxes bounds (bound_relax_factor = 1e-8: relaxed_bounds; states in that slack are
    projected back at the end, IPOPT's honor_original_bounds), so an initial state ON a state bound is inside the box (the reference's NLP does not bound x_0 at all, PDP.py:141-146; its barrier term here is a finite constant).  This is the classical primal
    barrier method (Fiacco & McCormick), not IPOPT's primal-dual iteration: the same solution (to O(mu_min) = tol / 10), a different path to it.
    Returns the dict of solve_batch (cost = the ORIGINAL objective along the returned trajectory; costate = multipliers of the dynamics = IPOPT's lam_g) plus
    "kernel_converged" [B] (the last barrier sub-problem met the kernel's own test) beside "converged" (that, or the documented fp64 floor of the primal barrier)."""
    if unsupported:
        raise NotImplementedError("bounded ocSolver: %s not supported with finite bounds (only tol, max_iter, print_level)" % ", ".join(sorted(unsupported)))
    torch = runtime.torch_cuda()
    mdl = oc.model()
    n, m, T = mdl.n, mdl.m, int(horizon)
    if n > 16 or m > 4:
        raise NotImplementedError("bounded ocSolver: the multiple-shooting kernel serves n <= 16, m <= 4")
    x0 = runtime.dev(ini_state).reshape(-1, n)
    B = x0.shape[0]
    th = np.asarray(oc._theta(auxvar_value, B), dtype=np.float64).reshape(-1, oc.n_auxvar)
    x0n = x0.cpu().numpy()
    lbx, ubx, lbu, ubu = relaxed_bounds(oc, x0n)
    if bool(((x0n <= lbx[None]) | (x0n >= ubx[None])).any()):
        # the reference's NLP does not apply the state bounds to x_0 (PDP.py:144-146), so IPOPT would accept this; in the barrier sub-problems the term of x_0 is a
        # constant that must be finite: on the bound it is (relaxed bounds), beyond it is not
        raise NotImplementedError("bounded ocSolver: an initial state outside the state bounds is not supported")
    bar = oc.barrier_model((lbx, ubx, lbu, ubu))

    def push(z, lb, ub):                    # IPOPT's projection of the starting point into the interior of [lb, ub]
        lo = np.where(np.abs(lb) < 1e19, lb + np.minimum(1e-2 * np.maximum(1.0, np.abs(lb)), 1e-2 * (ub - lb)), -np.inf)
        hi = np.where(np.abs(ub) < 1e19, ub - np.minimum(1e-2 * np.maximum(1.0, np.abs(ub)), 1e-2 * (ub - lb)), np.inf)
        return np.minimum(np.maximum(z, lo), hi)

    def w0(lb, ub):                         # the reference's initial guess 0.5 (lb + ub) (PDP.py:155,167: 0 for the +-1e20 defaults; for a one-sided bound 0 as well - IPOPT pushes it)
        both = (np.abs(lb) < 1e19) & (np.abs(ub) < 1e19)
        return np.where(both, 0.5 * (lb + ub), 0.0)
    xg = np.tile(push(w0(lbx, ubx), lbx, ubx), (B, T + 1, 1))
    xg[:, 0] = x0n
    ug = np.tile(push(w0(lbu, ubu), lbu, ubu), (B, T, 1))
    warm = (runtime.dev(xg), runtime.dev(ug), torch.zeros((B, T, n), dtype=torch.float64, device="cuda"))
    mu_min, mu, iters = max(0.1 * tol, 1e-9), float(mu0), 0
    accepted = None
    while True:
        th2 = np.concatenate([np.broadcast_to(th, (B if th.shape[0] == B else 1, th.shape[1])), np.full((B if th.shape[0] == B else 1, 1), mu)], axis=1)
        sub_tol = max(0.1 * tol, min(1e-4, mu))           # sub-problems only as accurately as their mu deserves (IPOPT: E_mu <= 10 mu)
        ms = bar.oc_solve_ms(x0, th2 if th2.shape[0] == B and B > 1 else th2[0], T, tol=sub_tol, max_iter=max_iter, warm=warm)
        iters += int(ms["iterations"].max())
        # A sub-problem counts as solved when the kernel says so, or when its line search has nothing left to gain at a feasible point whose stationarity
        # residual is at the floor this method has in fp64: the barrier gradient mu / (bound - v) is formed from a slack that cancels (v within mu / z of
        # its bound: relative error eps |bound| z / mu in the slack), so below mu ~ 1e-7 the residual stalls around 1e-6 - 1e-5 (IPOPT carries the bound
        # multipliers as variables of their own and does not have that floor; the trajectory itself keeps converging: cost and controls agree with an
        # independent solution to 1e-9 / 1e-6, tests/test_gpu_ocsolver.py)
        lam_scale = 1.0 + ms["costate"].abs().amax(dim=(1, 2))
        x_scale = 1.0 + torch.maximum(ms["state"].abs().amax(dim=(1, 2)), ms["control"].abs().amax(dim=(1, 2)))
        floor_ok = (ms["resid"][:, 0] <= tol * x_scale) & (ms["resid"][:, 1] <= 1e-4 * lam_scale) & torch.isfinite(ms["cost"])
        accepted = ms["converged"] | (floor_ok & (mu < 1e-6))
        if print_level > 0:
            print("  barrier mu = %.2e: %d/%d converged (%d accepted), max %d iterations" % (mu, int(ms["converged"].sum()), B, int(accepted.sum()), int(ms["iterations"].max())))
        warm = (ms["state"], ms["control"], ms["costate"])
        if mu <= mu_min:
            break
        mu = max(mu_min, min(0.2 * mu, mu ** 1.5))
    # (where a state bound was relaxed for an initial state on it, later states may sit in the 1e-8 slack: projected back, IPOPT's honor_original_bounds)
    f64 = dict(dtype=torch.float64, device="cuda")
    lo_x, hi_x = torch.as_tensor(np.asarray(oc.state_lb, float), **f64), torch.as_tensor(np.asarray(oc.state_ub, float), **f64)
    ms["state"][:, 1:] = torch.minimum(torch.maximum(ms["state"][:, 1:], lo_x), hi_x)
    xr, cost = mdl.oc_rollout(x0, ms["control"], oc._theta(auxvar_value, B))
    feas = (xr - ms["state"]).abs().amax(dim=(1, 2)) <= 1e3 * tol * (1 + ms["state"].abs().amax(dim=(1, 2)))
    return {"state": ms["state"], "control": ms["control"], "costate": ms["costate"], "cost": cost, "grad_norm": ms["resid"][:, 1].contiguous(),
            "converged": accepted & feas, "kernel_converged": ms["converged"] & feas, "iterations": iters, "method_ms": ms["converged"].clone(), "status": ms["status"],
            "barrier_mu": mu}
