"""GPU: the BASELINE.json configurations at (or near) full size, checked through size-independent properties the domain
offers: zero loss / zero gradient at the generating parameter, PDP gradient == finite differences of the re-solved
problem, per-sample results independent of batch composition, linearity of the sensitivity recursion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def npy(t):
    return t.detach().cpu().numpy()


def make_oc(name):
    from test_gpu_ocsolver import make_oc as mk
    return mk(name)


def test_C2_cartpole_irl_batch256_pdp_gradient_is_derivative_through_the_oc_solution():
    """C2: cart-pole IRL, n=4 m=1 T=50, 256 cost/dynamics parameter samples.  Demo = optimum at theta*; for every sample
    theta_b the OC problem is solved on the GPU and differentiated by PDP.  Properties: (i) at theta* loss = 0 and gradient = 0;
    (ii) PDP's gradient equals the central finite difference of loss(theta) obtained by RE-SOLVING the OC problem
    (the reference's dp is half the gradient of its loss, cartpole_PDP.py:63-74)."""
    from pdp_amd import ocsolver
    import torch
    oc = make_oc("cartpole")
    rng = np.random.default_rng(0)
    B, T = 256, 50
    th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
    x0 = np.zeros((B, 4))
    x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    demo = ocsolver.solve_batch(oc, x0, T, th_star, want_gains=True)
    assert bool(demo["converged"].all())
    out = oc.pdp_grad_batch(demo["control"], th_star, demo["state"], demo["control"], state_traj=demo["state"], costate_traj=demo["costate"])
    assert float(out["loss"].abs().max()) == 0.0 and float(out["grad"].abs().max()) == 0.0
    theta = th_star[None, :] + rng.uniform(-0.1, 0.1, (B, 7))
    sol = ocsolver.solve_batch(oc, x0, T, theta, warm_start=demo, want_gains=True)      # closed-loop warm start from the theta* solution
    assert bool(sol["converged"].all())
    out = oc.pdp_grad_batch(sol["control"], theta, demo["state"], demo["control"], state_traj=sol["state"], costate_traj=sol["costate"])
    assert int(out["status"].sum()) == 0
    g = npy(out["grad"])
    # finite differences through the solver, 8 samples x 7 parameters solved as one batch of 112 problems
    idx = np.arange(8)
    eps = 1e-5
    thp = np.repeat(theta[idx], 14, axis=0)
    for k in range(7):
        thp[k::14][:, k] += eps                     # rows 0..6: +eps on parameter k
        thp[7 + k::14][:, k] -= eps                 # rows 7..13: -eps
    rep = lambda a: torch.repeat_interleave(a[:8], 14, dim=0)
    s2 = ocsolver.solve_batch(oc, np.repeat(x0[idx], 14, axis=0), T, thp, tol=1e-11,
                              warm_start={k: rep(sol[k]) for k in ("state", "control", "gains")})
    dx = s2["state"] - rep(demo["state"])
    du = s2["control"] - rep(demo["control"])
    L = npy((dx ** 2).sum(dim=(1, 2)) + (du ** 2).sum(dim=(1, 2))).reshape(8, 14)
    fd = (L[:, :7] - L[:, 7:]) / (2 * eps)
    assert np.abs(2 * g[idx] - fd).max() <= 2e-4 * np.abs(fd).max()


def test_C3_quadrotor_planning_T50_batch1024(margins):
    """C3's ControlPlanning half at full size: quadrotor n=13 m=4, Lagrange policy p=24, T=50, 1024 random initial states, shared theta
    (bench.py's workload).  Four samples against the oracle (PDP.py:850-878 restated, oracle/pdp_oracle.py) with recorded margins; all samples:
    finite, independent of the batch they are in; the gradient is the derivative of the rollout cost (central differences); the materialised
    route of the reference (integrateSys -> getAuxSys -> integrateAuxSys -> chain rule) gives the same numbers."""
    from oracle import models, pdp_oracle as po
    from pdp_amd import runtime as rt, zoo
    mdl = zoo.get("quadrotor", "oc")
    rng = np.random.default_rng(0)
    B, T, p = 1024, 50, 24
    x0 = np.zeros((B, 13))
    x0[:, :3] = rng.uniform(-5, 5, (B, 3))
    x0[:, 6] = 1
    theta = rng.standard_normal(p)
    pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    loss, grad, x, u = mdl.cp_step(pol, p, x0, theta, T, want_traj=True)
    L, G = npy(loss), npy(grad)
    assert np.all(np.isfinite(L)) and np.all(np.isfinite(G))
    cp = po.make_cp(models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1), 0.1)
    cp.init_step(T)
    for i in (0, 341, 682, 1023):
        l, g = cp.step(x0[i], T, theta)
        sol = cp.integrateSys(x0[i], T, theta)
        margins.check("C3 ControlPlanning.step B=1024 sample %d vs oracle: loss (relative)" % i, abs(L[i] - l) / abs(l), 1e-11)
        margins.check("C3 ControlPlanning.step B=1024 sample %d vs oracle: gradient (relative to its largest entry)" % i, np.abs(G[i] - g).max() / np.abs(g).max(), 1e-10)
        margins.check("C3 ControlPlanning.step B=1024 sample %d vs oracle: state trajectory (absolute)" % i, np.abs(npy(x)[i] - sol["state_traj"]).max(), 1e-10)
    l2, g2 = mdl.cp_step(pol, p, x0[500:503], np.tile(theta, (3, 1)), T)
    assert np.array_equal(npy(l2), L[500:503]) and np.array_equal(npy(g2), G[500:503])
    l3, g3 = mdl.cp_step_materialised(pol, p, x0[:64], theta, T)
    assert np.abs(npy(l3) - L[:64]).max() <= 1e-12 * np.abs(L[:64]).max() and np.abs(npy(g3) - G[:64]).max() <= 1e-10 * np.abs(G[:64]).max()
    eps = 1e-6
    for k in (0, 11, 23):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        lp, _ = mdl.cp_step(pol, p, x0[:8], tp, T)
        lm, _ = mdl.cp_step(pol, p, x0[:8], tm, T)
        fd = (npy(lp) - npy(lm)) / (2 * eps)
        # (central differences of a cost of size |L| carry ~ eps_machine |L| / eps of rounding)
        assert np.abs(fd - G[:8, k]).max() <= 1e-6 * np.abs(G[:8, k]).max() + 20 * 2.2e-16 * np.abs(L[:8]).max() / eps


def test_C4_rocket_planning_T100_batch512():
    """C4 per-GPU shard: rocket n=13 m=3 T=100, Lagrange policy p=18, 512 random initial states.  ControlPlanning.step returns
    the exact gradient of the rollout cost: checked by finite differences; per-sample results do not depend on the batch."""
    from pdp_amd import runtime as rt, zoo
    from pdp_amd import JinEnv
    mdl = zoo.get("rocket", "oc")
    rng = np.random.default_rng(1)
    B, T, p = 512, 100, 18
    x0 = np.zeros((B, 13))
    x0[:, 0:3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
    theta = 0.5 * rng.standard_normal(p)
    pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    loss, grad = mdl.cp_step(pol, p, x0, theta, T)
    loss, grad = npy(loss), npy(grad)
    assert np.all(np.isfinite(loss)) and np.all(np.isfinite(grad))
    l2, g2 = mdl.cp_step(pol, p, x0[100:103], np.tile(theta, (3, 1)), T)
    assert np.array_equal(npy(l2), loss[100:103]) and np.array_equal(npy(g2), grad[100:103])
    eps = 1e-6
    for k in (0, 7, 17):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        lp, _ = mdl.cp_step(pol, p, x0[:4], tp, T)
        lm, _ = mdl.cp_step(pol, p, x0[:4], tm, T)
        fd = (npy(lp) - npy(lm)) / (2 * eps)
        assert np.abs(fd - grad[:4, k]).max() <= 1e-6 * np.abs(grad[:4, k]).max()


C4_UNCONVERGED_BOUND = 0.01      # measured in round 6: 512 of 512 of this batch converge (profiles/r06_parity_margins.txt); bench.py's harder C4 batch: 383 of 512 (profiles/r06_solver_robustness.json)


def test_C4_rocket_fused_oc_unit_T100_p10_batch512(margins):
    """C4's OC unit on one GPU's shard: rocket n=13 m=3 p=10, T=100, B=512 through the fused kernel (rollout -> costates -> aux system in
    LDS -> Riccati -> gradient) at the full horizon.  Inputs: the optimal controls of 512 landing problems (solved on the GPU at T=100),
    perturbed by 2 %, at per-sample parameters perturbed by 5 %.  Four samples are compared with the oracle (trajectory, costates, and the
    sensitivities / loss / gradient against the 40-digit evaluation of the reference formulas on the kernel's own trajectory);
    all samples: clean status, results independent of the batch they are in."""
    from oracle import pdp_oracle as po
    from test_gpu_models import oracle_oc, rel, TOL
    from pdp_amd import zoo
    mdl = zoo.get("rocket", "irl")
    oc = oracle_oc("rocket")
    rng = np.random.default_rng(4)
    B, T = 512, 100
    th_star = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    x0 = np.zeros((B, 13))
    x0[:, 0:3] = np.array([10, -8, 5.0]) + 0.5 * rng.standard_normal((B, 3))
    x0[:, 3] = -0.5
    ang = 0.5 + 0.1 * rng.standard_normal(B)
    x0[:, 6], x0[:, 8], x0[:, 9] = np.cos(ang / 2), np.sin(ang / 2) / np.sqrt(2), -np.sin(ang / 2) / np.sqrt(2)
    sol = mdl.oc_solve_ms(x0, th_star, T)
    good = npy(sol["converged"])
    # cold-solve convergence of THIS batch within the kernel's 300 iterations, held to the measured rate (round-5 verdict, item 6: "good.sum() >= 0.9 * B" asserted loosely what
    # was recorded nowhere).  profiles/r06_solver_robustness.json has the same count for bench.py's C4 batch (383 / 512; 511 / 512 with IPOPT's own budget of 3000 iterations)
    margins.check("C4 rocket T=100 B=512 (this test's batch): cold solves NOT converged within 300 iterations, fraction", 1.0 - good.sum() / B, C4_UNCONVERGED_BOUND)
    demo_x, demo_u = npy(sol["state"]), npy(sol["control"])
    u = demo_u * (1 + 0.02 * rng.standard_normal(demo_u.shape))
    theta = th_star[None, :] * (1 + 0.05 * rng.standard_normal((B, 10)))
    out = mdl.oc_pdp_grad(u, theta, demo_x, demo_u, x0=x0, want_sens=True)
    st = npy(out["status"])
    assert int(st[good].sum()) == 0 and np.all(np.isfinite(npy(out["grad"])[good]))
    xg, lg = npy(out["x"]), npy(out["lam"])
    for i in np.nonzero(good)[0][[0, 1, 100, 400]]:
        xs = oc.rollout(x0[i], u[i], theta[i])
        assert rel(xg[i], xs) < 1e-9                      # T = 100 of unstable open-loop dynamics: 1-ulp differences of the products grow
        assert rel(lg[i], oc.costate(xg[i], u[i], theta[i])) < TOL
        aux = oc.getAuxSys(xg[i], u[i], lg[i], theta[i])
        ref64 = po.lqr_from_aux(aux, oc.n, oc.p, T)
        ex = po.lqr_solver_mp(aux["dynF"], aux["dynG"], aux["dynE"], aux["Hxx"], aux["Huu"], aux["Hxu"], aux["Hxe"], aux["Hue"],
                              aux["hxx"], aux["hxe"], np.zeros((oc.n, oc.p)), T)
        Xe, Ue = np.stack(ex["state_traj_opt"]), np.stack(ex["control_traj_opt"])
        tol_i = max(TOL, 2 * rel(np.stack(ref64["state_traj_opt"]), Xe))
        assert rel(npy(out["dxdp"])[i], Xe) < tol_i and rel(npy(out["dudp"])[i], Ue) < tol_i
        l, g = po.irl_loss_grad(xg[i], u[i], demo_x[i], demo_u[i], list(Xe), list(Ue))
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < tol_i
    sub = slice(200, 203)
    o2 = mdl.oc_pdp_grad(u[sub], theta[sub], demo_x[sub], demo_u[sub], x0=x0[sub])
    assert np.array_equal(npy(o2["grad"]), npy(out["grad"])[sub]) and np.array_equal(npy(o2["loss"]), npy(out["loss"])[sub])


def test_C5_quadrotor_sysid_T100_batch1024():
    """C5a per-GPU shard: quadrotor SysID T=100 p=5, 1024 trajectories generated at theta* with inputs U(-10,10):
    loss = 0 and gradient = 0 at theta*; away from it the (half-)gradient matches finite differences."""
    from pdp_amd import zoo
    from pdp_amd import JinEnv
    mdl = zoo.get("quadrotor", "sysid")
    rng = np.random.default_rng(2)
    B, T = 1024, 100
    th_star = np.array([1, 1, 1, 1, 0.4])
    u = rng.uniform(-1, 1, (B, T, 4)) + 2.5
    x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
    xobs = mdl.sysid_integrate(x0, u, th_star)
    loss, grad = mdl.sysid_step(u, xobs, th_star)
    assert float(loss.abs().max()) == 0.0 and float(grad.abs().max()) == 0.0
    theta = th_star + np.array([0.1, -0.05, 0.08, 0.03, -0.02])
    loss, grad = mdl.sysid_step(u, xobs, theta)
    g = npy(grad)
    eps = 1e-6
    for k in range(5):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        lp, _ = mdl.sysid_step(u[:8], xobs[:8], tp)
        lm, _ = mdl.sysid_step(u[:8], xobs[:8], tm)
        fd = (npy(lp) - npy(lm)) / (2 * eps)
        assert np.abs(fd - 2 * g[:8, k]).max() <= 1e-5 * max(np.abs(fd).max(), 1e-12)


def test_C5_quadrotor_neural_policy_T100_p420():
    """C5b: tanh-MLP policy hidden [13,13] (p = 420), T = 100: gradient of ControlPlanning.step vs finite differences on
    a few of the 420 weights (column-major layout), batch of 16."""
    from pdp_amd import runtime as rt, zoo
    from pdp_amd import JinEnv
    mdl = zoo.get("quadrotor", "oc")
    rng = np.random.default_rng(3)
    B, T, p = 16, 100, 420
    theta = 0.1 * rng.standard_normal(p)
    x0 = np.zeros((B, 13))
    x0[:, :3] = rng.uniform(-2, 2, (B, 3))
    x0[:, 6] = 1.0
    pol = rt.make_policy("mlp", layers=[13, 13, 4])
    loss, grad = mdl.cp_step(pol, p, x0, theta, T)
    g = npy(grad)
    assert g.shape == (B, p) and np.all(np.isfinite(g))
    eps = 1e-6
    for k in (0, 181, 200, 419):
        tp, tm = theta.copy(), theta.copy()
        tp[k] += eps
        tm[k] -= eps
        lp, _ = mdl.cp_step(pol, p, x0[:3], tp, T)
        lm, _ = mdl.cp_step(pol, p, x0[:3], tm, T)
        fd = (npy(lp) - npy(lm)) / (2 * eps)
        assert np.abs(fd - g[:3, k]).max() <= 1e-5 * max(np.abs(g[:3, k]).max(), 1e-9)


def test_C5_neural_policy_full_shard_equals_small_batch():
    """C5b at one GPU's shard (B = 1024: four trajectories per wavefront, cp_step_mlp4t_kernel) against a batch of 16 (one per wavefront, cp_step_mlp16_kernel;
    hidden activations of every time step in the HBM workspace): the arithmetic per trajectory does not depend on the batch or the kernel -
    loss and gradient of the first 16 trajectories agree to rounding with the 16-trajectory run."""
    from pdp_amd import runtime as rt, zoo
    mdl = zoo.get("quadrotor", "oc")
    rng = np.random.default_rng(3)
    B, T, p = 1024, 100, 420
    theta = 0.1 * rng.standard_normal(p)
    x0 = np.zeros((B, 13))
    x0[:, :3] = rng.uniform(-2, 2, (B, 3))
    x0[:, 6] = 1.0
    pol = rt.make_policy("mlp", layers=[13, 13, 4])
    # the larger of: one double per lane and time step (one trajectory per wavefront) | activations of four trajectories per wavefront in D layout + the trajectories
    wsb = lambda b: 8 * max(b * T * 64, ((b + 3) // 4) * T * 3 * 64 + b * ((T + 1) * 13 + T * 4))
    assert mdl.lib.pdp_cp_step_workspace_bytes(B, T, rt.C.byref(pol), p) == wsb(B)
    assert mdl.lib.pdp_cp_step_workspace_bytes(16, T, rt.C.byref(pol), p) == wsb(16)
    L, G = mdl.cp_step(pol, p, x0, theta, T)
    l, g = mdl.cp_step(pol, p, x0[:16], theta, T)
    L, G, l, g = npy(L), npy(G), npy(l), npy(g)
    assert np.all(np.isfinite(G)) and np.abs(L[:16] - l).max() <= 1e-13 * np.abs(l).max()
    assert np.abs(G[:16] - g).max() <= 1e-12 * np.abs(g).max()


def test_lqr_sensitivity_is_linear_in_the_parameter_columns():
    """size-independent property of the aux-system solve: the solution is linear in (E, Hxe, Hue, hxe, X0) - doubling those
    columns doubles X, U, Lambda; a zero right-hand side gives a zero solution (B = 2048 random problems)."""
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(4)
    B, T, n, m, p = 2048, 20, 13, 4, 9

    def spd(k, s, cnt):
        A = rng.standard_normal((cnt, k, k))
        return s * (A @ A.transpose(0, 2, 1) / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n))
    G = 0.3 * rng.standard_normal((B, T, n, m))
    Hxx, Huu, hxx = spd(n, 1.0, B * T).reshape(B, T, n, n), spd(m, 0.5, B * T).reshape(B, T, m, m), spd(n, 1.0, B)
    Hxu = 0.05 * rng.standard_normal((B, T, n, m))
    E, Hxe, Hue = 0.1 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
    hxe, X0 = 0.2 * rng.standard_normal((B, n, p)), rng.standard_normal((B, n, p))
    X1, U1, L1, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0)
    X2, U2, L2, _ = rt.lqr_solve(F, G, Hxx, Huu, hxx, 2 * hxe, E=2 * E, Hxu=Hxu, Hxe=2 * Hxe, Hue=2 * Hue, X0=2 * X0)
    assert int(st.sum()) == 0
    for a, b in ((X1, X2), (U1, U2), (L1, L2)):
        assert float((2 * a - b).abs().max()) <= 1e-12 * float(b.abs().max())
    X0_, U0_, L0_, _ = rt.lqr_solve(F, G, Hxx, Huu, hxx, 0 * hxe, Hxu=Hxu)
    assert float(X0_.abs().max()) == 0.0 and float(U0_.abs().max()) == 0.0
