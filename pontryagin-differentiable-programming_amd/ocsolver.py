"""Batched optimal-control solve on the GPU - stands where the reference's OCSys.ocSolver hands a multiple-shooting NLP to
IPOPT (PDP/PDP.py:121-220).  Stagewise Newton / iLQR on the single-shooting problem, every trajectory of the batch in
parallel, all arithmetic in the HIP kernels (Python only sequences launches and does the per-sample line-search bookkeeping):

    repeat:  costates  lambda = c_x + f_x' lambda+           (pdp_oc_costate_batched)
             F, G, Hxx, Hxu, Huu, hxx, H_u along (x,u,lambda) (pdp_oc_auxsys_batched)
             LQ sub-problem for (dx, du): the same Riccati kernel as LQR.lqrSolver with p = 1, Hue := H_u (pdp_lqr_solve_batched)
             closed-loop rollout u = ubar - alpha k - K (x - xbar), backtracking on the true cost (pdp_oc_rollout_feedback_batched)

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
    u_init (optional warm start) is used per sample only where its rollout is finite and cheaper than u = 0."""
    torch = runtime.torch_cuda()
    mdl = oc.model()
    n, m = mdl.n, mdl.m
    x0 = runtime.dev(ini_state).reshape(-1, n)
    B, T = x0.shape[0], int(horizon)
    th = oc._theta(auxvar_value, B)
    u = torch.zeros((B, T, m), dtype=torch.float64, device="cuda")
    x, J = mdl.oc_rollout(x0, u, th)
    if u_init is not None:
        uw = runtime.dev(u_init).reshape(B, T, m)
        xw, Jw = mdl.oc_rollout(x0, uw, th)
        better = torch.isfinite(Jw) & torch.isfinite(xw).all(dim=(1, 2)) & ((Jw < J) | ~torch.isfinite(J) | force_init)
        u, x, J = torch.where(better.view(B, 1, 1), uw, u), torch.where(better.view(B, 1, 1), xw, x), torch.where(better, Jw, J)
    zeros_lam = torch.zeros((B, T, n), dtype=torch.float64, device="cuda")
    hxe0 = torch.zeros((B, n, 1), dtype=torch.float64, device="cuda")
    mu = torch.zeros((B,), dtype=torch.float64, device="cuda")            # Levenberg-Marquardt damping per sample
    newton = torch.zeros((B,), dtype=torch.bool, device="cuda")            # per-sample mode: False = Gauss-Newton (iLQR), True = Newton
    eye_m = torch.eye(m, dtype=torch.float64, device="cuda")
    keys = ("dynF", "dynG", "Hxx", "Hxu", "Huu", "hxx", "dHu")
    converged = torch.zeros((B,), dtype=torch.bool, device="cuda")
    gnorm = torch.full((B,), float("inf"), dtype=torch.float64, device="cuda")
    it = 0
    last_nconv, last_gain = 0, 0
    for it in range(max_iter):
        lam = mdl.oc_costate(x, u, th)
        aux = mdl.oc_auxsys(x, u, lam, th, only=keys)
        gnorm = aux["dHu"].abs().amax(dim=(1, 2))
        scale = 1.0 + u.abs().amax(dim=(1, 2))
        converged = gnorm <= tol * scale
        if print_level:
            print("  ocsolver iter %3d  max|H_u| %.3e  mean cost %.10g  converged %d/%d  newton %d  max mu %.1e" % (
                it, float(gnorm.max()), float(J.mean()), int(converged.sum()), B, int(newton.sum()), float(mu.max())))
        if bool(converged.all()):
            break
        # stragglers: once most of the batch is done and nothing new has converged for a while, stop - solve_batch re-solves the
        # rest from the closed-loop warm start of a converged neighbour, which takes a handful of iterations
        nconv = int(converged.sum())
        if nconv > last_nconv:
            last_nconv, last_gain = nconv, it
        elif straggler_patience and nconv >= 0.9 * B and it - last_gain >= straggler_patience:
            break
        # Newton once the stationarity residual is small relative to the controls, Gauss-Newton (always a descent direction) before
        # (with hysteresis: a sample whose residual has grown back by two orders of magnitude is no longer in Newton's basin)
        newton = (newton & (gnorm <= 100.0 * NEWTON_SWITCH * scale)) | (gnorm <= NEWTON_SWITCH * scale)
        if bool((~newton).any()):                                            # Gauss-Newton Hessians = Hamiltonian Hessians at lambda = 0
            gn = mdl.oc_auxsys(x, u, zeros_lam, th, only=("Hxx", "Hxu", "Huu"))
            sel = (~newton).view(B, 1, 1, 1)
            Hxx, Hxu, Huu = torch.where(sel, gn["Hxx"], aux["Hxx"]), torch.where(sel, gn["Hxu"], aux["Hxu"]), torch.where(sel, gn["Huu"], aux["Huu"])
        else:
            Hxx, Hxu, Huu = aux["Hxx"], aux["Hxu"], aux["Huu"]
        Huu = Huu + mu.view(B, 1, 1, 1) * eye_m
        _, dU, _, status, gains = runtime.lqr_solve(aux["dynF"], aux["dynG"], Hxx, Huu, aux["hxx"], hxe0, Hxu=Hxu, Hue=aux["dHu"].unsqueeze(-1),
                                                    want_costate=False, return_gains=True)
        # first-order change of the cost along the open-loop direction, sum_t H_u' du, must be negative
        slope = (aux["dHu"] * dU.squeeze(-1)).sum(dim=(1, 2))
        bad = (status != 0) | ~torch.isfinite(dU).all(dim=(1, 2, 3)) | ~(slope < 0)
        alpha = torch.where(bad | converged, torch.zeros_like(J), torch.ones_like(J))
        accepted = converged | bad
        x_new, u_new, J_new = x, u, J
        for _ls in range(10):
            xt, ut, Jt = mdl.oc_rollout_feedback(x0, u, x, gains, alpha, th)
            ok = torch.isfinite(Jt) & (Jt <= J + 1e-4 * alpha * slope + 1e-13 * J.abs()) & ~accepted      # Armijo, up to rounding of J
            sel3 = ok.view(B, 1, 1)
            x_new, u_new, J_new = torch.where(sel3, xt, x_new), torch.where(sel3, ut, u_new), torch.where(ok, Jt, J_new)
            accepted = accepted | ok
            if bool(accepted.all()):
                break
            alpha = torch.where(accepted, alpha, alpha * 0.5)
        failed = (~accepted | bad) & ~converged                               # no acceptable step
        x, u, J = x_new, u_new, J_new                                         # (unchanged where nothing was accepted)
        # failure: Newton falls back to Gauss-Newton; Gauss-Newton raises its damping.  success: relax the damping.
        mu = torch.where(failed & ~newton, torch.clamp(mu * 4.0, min=1e-4), torch.where(failed, mu, mu * 0.5))
        mu = torch.where(mu < 1e-8, torch.zeros_like(mu), mu)
        newton = newton & ~failed
    lam = mdl.oc_costate(x, u, th)
    out = {"state": x, "control": u, "costate": lam, "cost": J, "grad_norm": gnorm, "iterations": it, "converged": converged}
    if want_gains:      # time-varying LQR feedback around the final trajectory (closed-loop warm starts of neighbouring problems)
        aux = mdl.oc_auxsys(x, u, lam, th, only=keys)
        _, _, _, _, gains = runtime.lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], hxe0, Hxu=aux["Hxu"],
                                              Hue=aux["dHu"].unsqueeze(-1), want_costate=False, return_gains=True)
        out["gains"] = gains.clone()
    return out


def solve_batch(oc, ini_state, horizon, auxvar_value, u_init=None, tol=1e-9, max_iter=300, print_level=0, neighbor_retries=2,
                warm_start=None, want_gains=False):
    """Batched OC solve (see _solve) plus a batch-level globalisation: samples that did not converge (non-convex problems such
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
                 want_gains=want_gains, straggler_patience=10 if (nb > 1 and neighbor_retries > 0) else 0)
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
