"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) - numpy restatement of what `OCSys.ocSolver` does with the multiple-shooting
NLP it builds (reference PDP/PDP.py:131-182): hand it to IPOPT with an all-zero initial guess (PDP.py:155,166; the state and
control bounds default to +-1e20, PDP.py:47-57,81-93, which IPOPT treats as "no bound", so the problem is equality-constrained
and the barrier machinery is idle).

The algorithm lives in a third-party dependency that is absent from /root/reference: IPOPT 3.x behind CasADi's `nlpsol('solver',
'ipopt', ...)` (the reference pins neither; CasADi 3.5.x wheels ship IPOPT 3.12/3.14 with MUMPS).  What is restated here is its
PUBLISHED algorithm - A. Waechter, L. T. Biegler, "On the implementation of an interior-point filter line-search algorithm for
large-scale nonlinear programming", Math. Program. 106 (2006) - specialised to the equality-constrained case, default options:

  * primal-dual Newton step on the KKT system  [W + dw I, A'; A, 0] [d; dlam] = -[grad f + A' lam; c]          (eq. 13 / 26)
  * inertia correction: dw = 0 first; on wrong inertia dw = 1e-4 (or dw_last / 3), then x100 (x8 once a correction has been
    used before) until the reduced Hessian is positive definite                                                 (Algorithm IC)
  * filter line search on theta = ||c||_1 and phi = f: switching condition (19) with s_phi = 2.3, s_theta = 1.1, delta = 1,
    Armijo condition (20) with eta = 1e-8, sufficient decrease (18) with gamma_theta = 1e-5, gamma_phi = 1e-8, theta_min /
    theta_max = 1e-4 / 1e4 x max(1, theta(x0)), filter augmented after every non-f-type step, step halving     (Algorithm A)
  * equality multipliers move with the primal step length, lam+ = lam + alpha dlam; initial multipliers from the least-squares
    estimate, set to zero when its max-norm exceeds 1000                                                         (section 3.6)
  * OPTIONAL (solve(soc=True); round 5): second-order correction (section 2.4, steps A-5.5 .. A-5.10; max_soc = 4, kappa_soc = 0.99):
    when the FIRST trial point of an iteration is rejected and its constraint violation is not below the iterate's, the step is
    corrected by solving the same KKT matrix with the constraint block c_soc = alpha c(x_k) + c(x_k + alpha d) (accumulated over the
    attempts); the corrected point is tested with the original alpha and directional derivative.  IPOPT has it on by default, and
    cold solves do take corrected steps (cart-pole demo 0: 22 tried, 7 taken, 35 iterations instead of 45) - rounds 1-4 claimed the
    opposite.  It is nevertheless OFF by default here, on the evidence of the reference's own data: every stored optimum is reached
    either way, but cart-pole demonstration 4 at the first row of the stored IRL trace - a non-convex solve, inertia corrections at
    most iterations - ends WITH the correction as published in another stationary point (cost 1513.67 after 2490 iterations) where
    the reference's IPOPT run (stored loss_trace[1]) and this restatement without it end in 623.79 (tests/test_oracle_soc.py).  The
    published algorithm is not all of IPOPT (watchdog, its restoration NLP): as a model of what the reference's solver RETURNED the
    restatement is better without the correction.
Not restated: watchdog.  Restoration phase: IPOPT's own restoration algorithm is not restated; a step that would enter it (one stored demo: robot arm 3)
is answered by the structure-specific feasibility restoration described in `solve` (states <- rollout of the controls), which satisfies what the
filter method asks of a restoration phase and lands in IPOPT's stored optimum on that demo; `restoration=False` raises there instead.

The KKT system is solved stage by stage: the Newton step is the solution of an LQ problem with affine terms (defects c_t in the
dynamics, Lagrangian gradients in the cost), i.e. the same backward Riccati / forward rollout as `LQR.lqrSolver`
(PDP.py:557-608) with one "parameter" column, and its inertia is correct iff every Quu_t of the recursion is positive definite.
That is also how the HIP kernel (csrc: oc_solve_ms_kernel) is organised, so the per-iteration log of this file is what the
kernel is debugged against.

Pinned: reproduces the reference's stored IPOPT optima (tests/golden/demos_*.npz: state, control, lam_g, cost) from the zero
guess on all five systems, and the stored IRL loss traces (irltrace_*.npz) at the reference's iterates - tests/test_oracle_golden.py.
"""
import numpy as np

from .pdp_oracle import _vec

# IPOPT default options used by the restatement (option names of the IPOPT documentation)
OPT = dict(s_phi=2.3, s_theta=1.1, delta=1.0, eta_phi=1e-8, gamma_theta=1e-5, gamma_phi=1e-8, theta_min_fact=1e-4, theta_max_fact=1e4,
           first_hessian_perturbation=1e-4, min_hessian_perturbation=1e-20, max_hessian_perturbation=1e20,
           perturb_inc_fact_first=100.0, perturb_inc_fact=8.0, perturb_dec_fact=1.0 / 3.0, constr_mult_init_max=1000.0,
           alpha_red_factor=0.5, alpha_min_frac=0.05, max_soc=4, kappa_soc=0.99)


def evaluate(oc, xs, us, lam, e):
    """Everything one iteration needs at (x, u, lam): per-stage F, G, Hessians of the Hamiltonian at lam_t (= multiplier of
    f(x_t,u_t) - x_{t+1}), defects, gradients of the Lagrangian, objective and constraint violation."""
    n, m, T = oc.n, oc.m, us.shape[0]
    ev = dict(F=[], G=[], Hxx=[], Hxu=[], Huu=[], c=np.zeros((T, n)), rdx=np.zeros((T + 1, n)), rdu=np.zeros((T, m)))
    f = 0.0
    for t in range(T):
        x, u, l = xs[t], us[t], lam[t]
        ev["F"].append(oc._m(oc.dfx_fn(x, u, e), n, n))
        ev["G"].append(oc._m(oc.dfu_fn(x, u, e), n, m))
        ev["Hxx"].append(oc._m(oc.ddHxx_fn(x, u, l, e), n, n))
        ev["Hxu"].append(oc._m(oc.ddHxu_fn(x, u, l, e), n, m))
        ev["Huu"].append(oc._m(oc.ddHuu_fn(x, u, l, e), m, m))
        ev["c"][t] = _vec(oc.dyn_fn(x, u, e)) - xs[t + 1]
        ev["rdu"][t] = _vec(oc.dHu_fn(x, u, l, e))
        if t >= 1:                                   # x_0 is fixed (lbw = ubw = ini_state, PDP.py:141-144): no stationarity row
            ev["rdx"][t] = _vec(oc.dHx_fn(x, u, l, e)) - lam[t - 1]
        f += float(oc.path_cost_fn(x, u, e))
    ev["rdx"][T] = _vec(oc.dhx_fn(xs[T], e)) - lam[T - 1]
    ev["hxx"] = oc._m(oc.ddhxx_fn(xs[T], e), n, n)
    ev["f"] = f + float(oc.final_cost_fn(xs[T], e))
    ev["theta"] = float(np.abs(ev["c"]).sum())
    ev["inf_pr"] = float(np.abs(ev["c"]).max())
    ev["inf_du"] = float(max(np.abs(ev["rdx"]).max(), np.abs(ev["rdu"]).max()))
    return ev


def objective_and_violation(oc, xs, us, e, defects=False):
    T = us.shape[0]
    c = np.stack([_vec(oc.dyn_fn(xs[t], us[t], e)) - xs[t + 1] for t in range(T)])
    if defects:
        return oc.cost(xs, us, e), float(np.abs(c).sum()), c
    return oc.cost(xs, us, e), float(np.abs(c).sum())


def kkt_step(ev, dw, n, m, c=None):
    """Newton step of the KKT system by the stage-wise recursion.  Returns (dx [T+1,n], du [T,m], dlam [T,n], inertia_ok).
    c [T, n]: constraint block of the right-hand side instead of the defects of `ev` (the second-order correction)."""
    T = len(ev["F"])
    if c is not None:
        ev = dict(ev)
        ev["c"] = c
    P = ev["hxx"] + dw * np.eye(n)
    W = ev["rdx"][T].copy()
    Ks, ks, Ps, Ws = [None] * T, [None] * T, [None] * (T + 1), [None] * (T + 1)
    Ps[T], Ws[T] = P, W
    ok = True
    for t in range(T - 1, -1, -1):
        F, G = ev["F"][t], ev["G"][t]
        Pc = P @ ev["c"][t] + W
        Quu = ev["Huu"][t] + dw * np.eye(m) + G.T @ P @ G
        Qux = ev["Hxu"][t].T + G.T @ P @ F
        Qu = ev["rdu"][t] + G.T @ Pc
        Qxx = ev["Hxx"][t] + dw * np.eye(n) + F.T @ P @ F
        Qx = ev["rdx"][t] + F.T @ Pc
        Quu = 0.5 * (Quu + Quu.T)
        if not np.all(np.isfinite(Quu)) or np.linalg.eigvalsh(Quu).min() <= 0.0:
            return None, None, None, False
        Ks[t] = np.linalg.solve(Quu, Qux)
        ks[t] = np.linalg.solve(Quu, Qu)
        P = Qxx - Qux.T @ Ks[t]
        P = 0.5 * (P + P.T)
        W = Qx - Qux.T @ ks[t]
        Ps[t], Ws[t] = P, W
    dx, du, dl = np.zeros((T + 1, n)), np.zeros((T, m)), np.zeros((T, n))
    for t in range(T):
        du[t] = -Ks[t] @ dx[t] - ks[t]
        dx[t + 1] = ev["F"][t] @ dx[t] + ev["G"][t] @ du[t] + ev["c"][t]
        dl[t] = Ps[t + 1] @ dx[t + 1] + Ws[t + 1]
    return dx, du, dl, ok


def least_squares_multipliers(ev, n, m):
    """min ||grad f + A' lam||: [I A'; A 0][w; lam] = -[grad f; 0] - the same recursion with W = I, no defects."""
    T = len(ev["F"])
    ls = dict(ev)
    ls["Hxx"] = [np.eye(n)] * T
    ls["Huu"] = [np.eye(m)] * T
    ls["Hxu"] = [np.zeros((n, m))] * T
    ls["hxx"] = np.eye(n)
    ls["c"] = np.zeros((T, n))
    _, _, dl, ok = kkt_step(ls, 0.0, n, m)
    return dl if ok else np.full((T, n), np.nan)          # (a non-finite point: the callers then start from zero multipliers, as for an estimate above constr_mult_init_max)


def solve(oc, ini_state, horizon, auxvar_value, tol=1e-10, max_iter=300, log=None, restoration=True, u_init=None, warm=None, soc=False, watchdog=False):
    """ocSolver's NLP (PDP.py:131-182) solved the way IPOPT does.  Returns the reference's result fields plus `iterations` and `restorations`.
    log: optional list receiving one dict per iteration (objective, inf_pr, inf_du, dw, alpha, step type).
    restoration: what happens when the line search falls below alpha_min, where IPOPT switches to its feasibility restoration phase.  IPOPT's own
    restoration algorithm (an interior-point solve of min ||c||_1 + zeta/2 ||D_R (x - x_R)||^2) is not restated; what the filter method requires of
    that phase is a point that is acceptable to the filter with a smaller constraint violation (Waechter & Biegler 2006, section 3.3), and the
    multiple-shooting structure offers one directly: keep the controls, replace the states by the ROLLOUT x_{t+1} = f(x_t, u_t) from the fixed x_0
    (theta = 0, acceptable to every filter entry).  As in IPOPT the current point is added to the filter first and the multipliers are reset to the
    least-squares estimate afterwards (zero if larger than constr_mult_reset_threshold = 1000).  False: raise instead (the pre-round-3 behaviour).
    u_init [T, m]: start from these controls and their rollout instead of the reference's all-zero guess (not something the reference does - it is
    the starting point PDP_MS_FROM_CONTROLS gives the kernel, restated here so that path has a checker).
    warm = (state [T+1, n], control [T, m], costate [T, n]): start the iteration AT that point (x_0 replaced by ini_state, no least-squares multiplier estimate) -
    what PDP_MS_WARM does in the kernel; with the point predict_start below returns, the start of an IRL loop's next solve.
    soc: second-order correction (PDP_MS_WITH_SOC in the kernel; the header says why it is off by default); the log rows carry `soc` = the number of corrections tried in
    the iteration and `soc_taken`; the result the total `soc_steps`.
    watchdog (round 6, EXPERIMENT - not in the kernels, parity unpinned: restated from the structure of IPOPT's BacktrackingLineSearch as remembered, no source or
    run of IPOPT to check it against): after watchdog_shortened_iter_trigger = 10 consecutive iterations whose accepted step was shortened the current iterate and
    direction are stored; for up to watchdog_trial_iter_max = 3 iterations the FULL step is taken whether acceptable or not, each tested against the stored point's
    (theta, phi, grad(phi)'d) and the filter; the first acceptable one ends the procedure, otherwise the stored iterate comes back and is searched along its stored
    direction from alpha = 1/2.  The log rows carry `wd` ("start", "trial", "success", "stop"); the result `watchdog_starts` / `watchdog_successes`.
    probes/watchdog_experiment.py asks what it does to the cold rocket solves at T = 100 that crawl with steps of 1e-3.  Checked once against the one stored case that
    tells mechanisms apart (cart-pole, first row of the stored IRL trace, tests/test_oracle_soc.py): the watchdog alone is never armed on those five solves and the stored
    loss is reproduced (3.8e-11); together with the second-order correction demonstration 4 still ends in 1513.67 where IPOPT's run ended in 623.79 - the restated PAIR is
    not what IPOPT ran, so neither default moved."""
    o = OPT
    e = _vec(auxvar_value)
    n, m, T = oc.n, oc.m, int(horizon)
    xs = np.zeros((T + 1, n))
    xs[0] = _vec(ini_state)
    us = np.zeros((T, m))                               # w0 = 0.5 (lb + ub) = 0   (PDP.py:155, 166)
    if u_init is not None:
        us = np.array(u_init, dtype=float).reshape(T, m)
        for t in range(T):
            xs[t + 1] = _vec(oc.dyn_fn(xs[t], us[t], e))
    lam = np.zeros((T, n))
    if warm is not None:
        assert u_init is None
        xs, us, lam = (np.array(a, dtype=float) for a in warm)
        xs[0] = _vec(ini_state)
        ev = evaluate(oc, xs, us, lam, e)
    else:
        ev = evaluate(oc, xs, us, lam, e)
        # initial multipliers: least-squares estimate from grad f (rdx / rdu at lam = 0)
        lam0 = least_squares_multipliers(ev, n, m)
        if np.all(np.isfinite(lam0)) and np.abs(lam0).max() <= o["constr_mult_init_max"]:
            lam = lam0
            ev = evaluate(oc, xs, us, lam, e)
    theta_max = o["theta_max_fact"] * max(1.0, ev["theta"])
    theta_min = o["theta_min_fact"] * max(1.0, ev["theta"])
    filt = []
    dw_last = 0.0
    it = 0
    n_rest = 0
    n_soc_total = 0
    in_wd, wd_short, wd_trial, wd, n_wd, n_wd_ok = False, 0, 0, None, 0, 0
    for it in range(max_iter + 1):
        f, theta = ev["f"], ev["theta"]
        inf_pr_it, inf_du_it = ev["inf_pr"], ev["inf_du"]
        scale = 1.0 + max(np.abs(xs).max(), np.abs(us).max())
        lscale = 1.0 + np.abs(lam).max()
        if ev["inf_pr"] <= tol * scale and ev["inf_du"] <= tol * lscale:
            break
        if it == max_iter:
            raise RuntimeError("ipopt_ms: no convergence in %d iterations" % max_iter)
        # ---- search direction with inertia correction (Algorithm IC)
        dw = 0.0
        wd_fell_back = False
        while True:
            dx, du, dl, ok = kkt_step(ev, dw, n, m)
            if ok:
                break
            if dw == 0.0:
                dw = o["first_hessian_perturbation"] if dw_last == 0.0 else max(o["min_hessian_perturbation"], o["perturb_dec_fact"] * dw_last)
            else:
                dw *= o["perturb_inc_fact_first"] if dw_last == 0.0 else o["perturb_inc_fact"]
            if dw > o["max_hessian_perturbation"]:
                if in_wd:
                    # a free watchdog trial landed where the iteration cannot go on (not finite, or no perturbation succeeds): the procedure ends, the stored iterate
                    # comes back and the iteration goes on from it as a regular one (direction computed anew); one log row and one iteration are spent
                    if log is not None:
                        log.append(dict(it=it, f=wd["f"], inf_pr=wd["ev"]["inf_pr"], inf_du=wd["ev"]["inf_du"], dw=dw, alpha=0.0, ftype=False, gd=wd["gd"], theta=wd["theta"],
                                        soc=0, soc_taken=False, wd="fallback"))
                    xs, us, lam, ev = wd["xs"], wd["us"], wd["lam"], wd["ev"]
                    in_wd, wd_short, wd_fell_back = False, 0, True
                    break
                raise RuntimeError("ipopt_ms: inertia correction failed")
        if wd_fell_back:
            continue
        if dw > 0.0:
            dw_last = dw
        # grad phi' d = rd' d + lam' c   (A d = -c)
        gd = float((ev["rdx"] * dx).sum() + (ev["rdu"] * du).sum() + (lam * ev["c"]).sum())
        # ---- backtracking filter line search (Algorithm A, steps A-5)
        alpha, accepted, ftype = 1.0, False, False
        if gd < 0.0:
            amin = min(o["gamma_theta"], o["gamma_phi"] * theta / (-gd))
            if theta <= theta_min:
                amin = min(amin, o["delta"] * theta ** o["s_theta"] / (-gd) ** o["s_phi"])
        else:
            amin = o["gamma_theta"]
        amin *= o["alpha_min_frac"]
        def acceptable(ft, tht, a, ref=None):
            """(accepted, f-type) of a trial point with objective ft and violation tht, tested with the step length a (steps A-5.3, A-5.4); ref = (f, theta, gd) of the
            point the test refers to (the iterate; in a watchdog procedure the stored point)"""
            f_, theta_, gd_ = (f, theta, gd) if ref is None else ref
            if not (np.isfinite(ft) and np.isfinite(tht) and tht <= theta_max and all(not (tht >= th_f and ft >= f_f) for th_f, f_f in filt)):
                return False, False
            switching = gd_ < 0.0 and a * (-gd_) ** o["s_phi"] > o["delta"] * theta_ ** o["s_theta"]
            if theta_ <= theta_min and switching:
                ok_ = ft <= f_ + o["eta_phi"] * a * gd_ + 10.0 * np.finfo(float).eps * abs(f_)
                return ok_, ok_
            return (tht <= (1.0 - o["gamma_theta"]) * theta_ or ft <= f_ - o["gamma_phi"] * theta_), False

        if watchdog and not in_wd and wd_short >= 10:
            in_wd, wd_trial, n_wd = True, 0, n_wd + 1
            wd = dict(xs=xs.copy(), us=us.copy(), lam=lam.copy(), ev=ev, dx=dx, du=du, dl=dl, gd=gd, f=f, theta=theta, dw=dw, amin=amin, started=True)
        if in_wd:
            ref = (wd["f"], wd["theta"], wd["gd"])
            xt, ut = xs + dx, us + du
            ft, tht, ct = objective_and_violation(oc, xt, ut, e, defects=True)
            ok_wd, ftype_wd = acceptable(ft, tht, 1.0, ref)
            fin_wd = bool(np.isfinite(ft) and np.isfinite(tht))
            tag = None
            if ok_wd:
                in_wd, wd_short, n_wd_ok, tag = False, 0, n_wd_ok + 1, "success"
                if not ftype_wd:
                    filt.append(((1.0 - o["gamma_theta"]) * ref[1], ref[0] - o["gamma_phi"] * ref[1]))
            else:
                wd_trial += 1
                if fin_wd and wd_trial <= 3:
                    tag = "start" if wd.pop("started", False) else "trial"
            if tag is not None:         # the full step is taken (acceptable, or one of the three free trials)
                wd.pop("started", None)
                if log is not None:
                    log.append(dict(it=it, f=f, inf_pr=ev["inf_pr"], inf_du=ev["inf_du"], dw=dw, alpha=1.0, ftype=False, gd=gd, theta=theta, soc=0, soc_taken=False, wd=tag))
                xs, us, lam = xt, ut, lam + dl
                ev = evaluate(oc, xs, us, lam, e)
                continue
            # the procedure failed: back to the stored iterate, regular backtracking along the stored direction, the full step known to fail
            xs, us, lam, ev = wd["xs"], wd["us"], wd["lam"], wd["ev"]
            dx, du, dl, gd, f, theta, dw, amin = wd["dx"], wd["du"], wd["dl"], wd["gd"], wd["f"], wd["theta"], wd["dw"], wd["amin"]
            inf_pr_it, inf_du_it = ev["inf_pr"], ev["inf_du"]
            in_wd, wd_short = False, 0
            alpha = o["alpha_red_factor"]
            if log is not None:
                log.append(dict(it=it, f=f, inf_pr=inf_pr_it, inf_du=inf_du_it, dw=dw, alpha=0.0, ftype=False, gd=gd, theta=theta, soc=0, soc_taken=False, wd="stop"))

        n_soc, soc_taken = 0, False
        dl_step = dl
        while alpha >= amin:
            xt, ut = xs + alpha * dx, us + alpha * du
            ft, tht, ct = objective_and_violation(oc, xt, ut, e, defects=True)
            accepted, ftype = acceptable(ft, tht, alpha)
            if accepted:
                break
            if soc and alpha == 1.0 and np.isfinite(ft) and np.isfinite(tht) and tht >= theta:
                # second-order correction (A-5.5 .. A-5.10): same matrix (same dw), constraint block c_soc; no bounds -> the corrected step is taken in full
                # (theta_old starts at the violation of the rejected full-step point, as IPOPT's implementation does - IpFilterLSAcceptor::TrySecondOrderCorrection;
                #  the paper's step A-5.6 writes theta(x_k), which is not larger: the difference can only show in whether a SECOND correction is tried)
                c_soc, a_soc, th_old = ev["c"].copy(), alpha, tht
                while n_soc < o["max_soc"]:
                    c_soc = a_soc * c_soc + ct
                    dxs, dus, dls, ok = kkt_step(ev, dw, n, m, c=c_soc)
                    if not ok or not (np.all(np.isfinite(dxs)) and np.all(np.isfinite(dus)) and np.all(np.isfinite(dls))):
                        break
                    n_soc += 1
                    a_soc = 1.0
                    xt2, ut2 = xs + a_soc * dxs, us + a_soc * dus
                    ft2, tht2, ct2 = objective_and_violation(oc, xt2, ut2, e, defects=True)
                    accepted, ftype = acceptable(ft2, tht2, alpha)
                    if accepted:
                        xt, ut, dl_step, soc_taken = xt2, ut2, dls, True
                        break
                    if not (np.isfinite(ft2) and np.isfinite(tht2)) or tht2 > o["kappa_soc"] * th_old:
                        break
                    th_old, ct = tht2, ct2
                if accepted:
                    break
            alpha *= o["alpha_red_factor"]
        n_soc_total += n_soc
        if not accepted:
            if not restoration:
                raise RuntimeError("ipopt_ms: line search would enter the restoration phase (not restated)")
            if theta == 0.0:
                raise RuntimeError("ipopt_ms: restoration phase called at a feasible point")
            filt.append(((1.0 - o["gamma_theta"]) * theta, f - o["gamma_phi"] * theta))
            xr = xs.copy()
            for t in range(T):
                xr[t + 1] = _vec(oc.dyn_fn(xr[t], us[t], e))
            if not np.all(np.isfinite(xr)) or not np.isfinite(oc.cost(xr, us, e)):
                raise RuntimeError("ipopt_ms: restoration phase failed (the rollout of the current controls is not finite)")
            xs = xr
            ev = evaluate(oc, xs, us, np.zeros((T, n)), e)
            lam0 = least_squares_multipliers(ev, n, m)
            lam = lam0 if (np.all(np.isfinite(lam0)) and np.abs(lam0).max() <= o["constr_mult_init_max"]) else np.zeros((T, n))
            ev = evaluate(oc, xs, us, lam, e)
            n_rest += 1
            wd_short = 0
            if log is not None:
                log.append(dict(it=it, f=f, inf_pr=inf_pr_it, inf_du=inf_du_it, dw=dw, alpha=0.0, ftype=False, gd=gd, theta=theta, restoration=True, soc=n_soc, soc_taken=False))
            continue
        if not ftype:
            filt.append(((1.0 - o["gamma_theta"]) * theta, f - o["gamma_phi"] * theta))
        wd_short = 0 if (alpha == 1.0 or soc_taken) else wd_short + 1          # consecutive iterations with a shortened step (the watchdog's trigger)
        if log is not None:
            log.append(dict(it=it, f=f, inf_pr=ev["inf_pr"], inf_du=ev["inf_du"], dw=dw, alpha=alpha, ftype=ftype, gd=gd, theta=theta,
                            dx=dx, du=du, dlam=dl, soc=n_soc, soc_taken=soc_taken))
        xs, us, lam = xt, ut, lam + alpha * dl_step             # (a corrected step is a full one: alpha = 1 there)
        ev = evaluate(oc, xs, us, lam, e)
    return {"state_traj_opt": xs, "control_traj_opt": us, "costate_traj_opt": lam, "cost": ev["f"], "iterations": it, "restorations": n_rest, "soc_steps": n_soc_total,
            "inf_pr": ev["inf_pr"], "inf_du": ev["inf_du"], "watchdog_starts": n_wd, "watchdog_successes": n_wd_ok}


def predict_start(oc, state, control, costate, auxvar_value, dtheta, with_costate=True):
    """First-order prediction of the optimal (x, u, lambda) at auxvar_value + dtheta from the solution at auxvar_value: the auxiliary control system of PDP
    (PDP.py:272-314 getAuxSys, 557-608 lqrSolver) IS that derivative - X_t = dx_t/dtheta, U_t = du_t/dtheta, Lambda_t = P_{t+1} X_{t+1} + W_{t+1} =
    dlambda_{t+1}/dtheta - so  (x, u, lambda) + (X, U, Lambda) dtheta  is what pdp_oc_predict_batched computes from the gradient unit's outputs.  An IRL
    loop (Examples/IRL/quadrotor/uav_PDP.py:52-62) solves at theta_k, differentiates there, moves theta by lr * gradient: started from this point instead of
    the previous solution, `solve(..., warm=...)` needs one Newton iteration fewer (the first iteration of a plain warm start only re-derives this step).
    with_costate=False: multipliers left as they are (the variant without the Riccati record)."""
    from .pdp_oracle import lqr_from_aux
    T = np.size(control, 0)
    d = _vec(dtheta)
    aux = oc.getAuxSys(state, control, costate, auxvar_value)
    lq = lqr_from_aux(aux, oc.n, oc.p, T)
    X, U, L = np.stack(lq["state_traj_opt"]), np.stack(lq["control_traj_opt"]), np.stack(lq["costate_traj_opt"])
    return state + X @ d, control + U @ d, (costate + L @ d) if with_costate else np.array(costate, dtype=float)


GUARD_TRUST = 0.02          # = PDP_MS_GUARD_TRUST of include/pdp_hip.h


def scaled_kkt_error(oc, xs, us, lam, auxvar_value, primal_only=False):
    """max(inf_pr / (1 + max|x|,|u|), inf_du / (1 + max|lam|)) - the two quantities of solve()'s convergence test, each over the scale it is tested against;
    primal_only: the first of them alone"""
    ev = evaluate(oc, xs, us, lam, _vec(auxvar_value))
    pr = ev["inf_pr"] / (1.0 + max(np.abs(xs).max(), np.abs(us).max()))
    return pr if primal_only else max(pr, ev["inf_du"] / (1.0 + np.abs(lam).max()))


def guarded_start(oc, ini_state, state, control, costate, auxvar_prev, dtheta, with_costate=True):
    """PDP_MS_PREDICT_GUARD restated: the first-order prediction (predict_start) of the solution at auxvar_prev + dtheta is kept only if its scaled KKT error AT the new
    parameter is finite and not larger than that of the previous solution itself (the plain warm start); both points carry the new initial state in row 0.
    A prediction of states and controls only (with_costate=False) is judged by the primal part alone: both candidates carry the same multipliers, so the dual
    residual says nothing about the prediction.  Returns ((x, u, lam) to start solve(..., warm=...) from, rejected: bool).
    Why: along the reference's stored rocket IRL run (Examples/IRL/rocket/data/PDP_results_trial_0.mat, rows 0 -> 1) the parameter moves by 1 % where the
    sensitivities are of order 1e2: the predicted point has ~50 times the KKT error of the point it was meant to improve, and the Newton iteration started there
    converges to another stationary point (loss 10289.86 where IPOPT - from the all-zero guess - stored 1301.24, which the plain warm start reproduces)."""
    th1 = _vec(auxvar_prev) + _vec(dtheta)
    plain = [np.array(a, dtype=float) for a in (state, control, costate)]
    plain[0][0] = _vec(ini_state)
    pred = [np.array(a, dtype=float) for a in predict_start(oc, state, control, costate, auxvar_prev, dtheta, with_costate=with_costate)]
    pred[0][0] = _vec(ini_state)
    # (PDP_MS_GUARD_TRUST: a correction below 2 % of max(1, |value|) in every state and control is trusted without the comparison)
    corr = max(np.abs((pred[0] - plain[0])[1:] / np.maximum(1.0, np.abs(plain[0][1:]))).max(), np.abs((pred[1] - plain[1]) / np.maximum(1.0, np.abs(plain[1]))).max())
    if corr <= GUARD_TRUST:
        return tuple(pred), False
    e_plain = scaled_kkt_error(oc, *plain, th1, primal_only=not with_costate)
    with np.errstate(all="ignore"):
        e_pred = scaled_kkt_error(oc, *pred, th1, primal_only=not with_costate) if all(np.all(np.isfinite(a)) for a in pred) else np.inf
    if np.isfinite(e_plain) and not (np.isfinite(e_pred) and e_pred <= e_plain):
        return tuple(plain), True
    return tuple(pred), False
