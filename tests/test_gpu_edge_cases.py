"""GPU: edge cases of the hot path - minimal and chunk-boundary horizons, single-trajectory batches, maximum tile sizes,
ragged SysID batches, non-finite inputs (status flags instead of exceptions inside the kernel), limit violations (error codes)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def npy(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("T", [1, 2, 3, 16, 17, 33, 64, 65, 120])
def test_fused_unit_any_horizon_matches_oracle(golden_dir, T):
    """odd / even horizons (the step loops run two steps per trip), T = 1, horizons around the chunk size of this model (64 lane-steps:
    one chunk, 64 + 1 -> two chunks of 33 and 32) and a long one (LDS staging grows with T): cart-pole, B = 3"""
    from oracle import models, pdp_oracle as po
    from pdp_amd import zoo
    mdl = zoo.get("cartpole", "irl")
    st = models.IRL_SETUP["cartpole"]
    oc = po.make_oc(models.REGISTRY["cartpole"](**st["kwargs"]), st["dt"])
    rng = np.random.default_rng(T)
    B = 3
    x0 = 0.2 * rng.standard_normal((B, 4))
    u = 0.5 * rng.standard_normal((B, T, 1))
    th = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0]) * (1 + 0.05 * rng.standard_normal((B, 7)))
    dx, du = rng.standard_normal((B, T + 1, 4)), rng.standard_normal((B, T, 1))
    out = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, want_sens=True)
    assert int(out["status"].sum()) == 0
    for i in range(B):
        xs = oc.rollout(x0[i], u[i], th[i])
        # open-loop rollout of an unstable system: ulp-level differences (device sin/cos vs numpy) grow exponentially with T
        assert rel(npy(out["x"])[i], xs) < (TOL if T <= 64 else 1e-8)
        lam = oc.costate(npy(out["x"])[i], u[i], th[i])
        assert rel(npy(out["lam"])[i], lam) < TOL
        aux = oc.getAuxSys(npy(out["x"])[i], u[i], npy(out["lam"])[i], th[i])
        ex = po.lqr_solver_mp(aux["dynF"], aux["dynG"], aux["dynE"], aux["Hxx"], aux["Huu"], aux["Hxu"], aux["Hxe"], aux["Hue"], aux["hxx"], aux["hxe"],
                              np.zeros((4, 7)), T)
        Xe, Ue = np.stack(ex["state_traj_opt"]), np.stack(ex["control_traj_opt"])
        ref64 = np.stack(po.lqr_from_aux(aux, 4, 7, T)["state_traj_opt"])
        tol = max(TOL, 2 * rel(ref64, Xe))
        assert rel(npy(out["dxdp"])[i], Xe) < tol and rel(npy(out["dudp"])[i], Ue) < tol
        l, g = po.irl_loss_grad(npy(out["x"])[i], u[i], dx[i], du[i], list(Xe), list(Ue))
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < tol


def test_single_trajectory_and_shared_vs_replicated_theta():
    from pdp_amd import zoo
    mdl = zoo.get("rocket", "irl")
    rng = np.random.default_rng(0)
    B, T = 5, 40
    x0 = np.zeros((B, 13)); x0[:, :3] = rng.standard_normal((B, 3)) + [10, -8, 5]; x0[:, 6] = 1
    u = np.tile([10.0, 0, 0], (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3))
    th = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    dx, du = np.zeros((B, T + 1, 13)), np.zeros((B, T, 3))
    a = mdl.oc_pdp_grad(u, th, dx, du, x0=x0)
    g_all, l_all = npy(a["grad"]).copy(), npy(a["loss"]).copy()
    b = mdl.oc_pdp_grad(u, np.tile(th, (B, 1)), dx, du, x0=x0)
    assert np.array_equal(npy(b["grad"]), g_all)
    c = mdl.oc_pdp_grad(u[2:3], th, dx[2:3], du[2:3], x0=x0[2:3])           # B = 1
    assert np.array_equal(npy(c["grad"])[0], g_all[2]) and npy(c["loss"])[0] == l_all[2]


def test_nonfinite_input_raises_status_flag_only_for_that_sample():
    from pdp_amd import zoo
    mdl = zoo.get("quadrotor", "irl")
    rng = np.random.default_rng(1)
    B, T = 4, 20
    x0 = np.zeros((B, 13)); x0[:, 6] = 1
    u = 2.5 + 0.1 * rng.standard_normal((B, T, 4))
    u[1, 3, 0] = np.nan
    th = np.array([1, 1, 1, 1, .4, 1, 1, 5, 1.0])
    dx = np.zeros((B, T + 1, 13)); dx[:, :, 6] = 1
    out = mdl.oc_pdp_grad(u, th, dx, np.full((B, T, 4), 2.5), x0=x0)
    st = npy(out["status"])
    assert st[1] & 1 and st[0] == 0 and st[2] == 0 and st[3] == 0
    assert np.all(np.isfinite(npy(out["grad"])[[0, 2, 3]]))


def test_lqr_maximum_sizes_and_limits():
    """n = 16, m = 4, p = 60 (= 64 - m) is the largest problem of ONE launch of pdp_lqr_solve_batched; more parameter columns are solved in
    column blocks by the class surface; n > 16 or m > 4 take the size-generic kernel (test below and tests/test_gpu_generic_sizes.py): no size is refused"""
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(2)
    n, m, p, T, B = 16, 4, 60, 5, 2

    def spd(k, s):
        A = rng.standard_normal((k, k))
        return s * (A @ A.T / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n)); G = 0.3 * rng.standard_normal((B, T, n, m)); E = 0.1 * rng.standard_normal((B, T, n, p))
    Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)]); Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
    Hxe, Hue = 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
    hxx, hxe = np.stack([spd(n, 1.0) for _ in range(B)]), 0.2 * rng.standard_normal((B, n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxe=Hxe, Hue=Hue)
    assert int(st.sum()) == 0
    Z = T * [np.zeros((n, m))]
    sol = po.lqr_solver(list(F[1]), list(G[1]), list(E[1]), list(Hxx[1]), list(Huu[1]), Z, list(Hxe[1]), list(Hue[1]), [hxx[1]], [hxe[1]], np.zeros((n, p)), T)
    assert rel(npy(X)[1], np.stack(sol["state_traj_opt"])) < TOL and rel(npy(Lam)[1], np.stack(sol["costate_traj_opt"])) < TOL
    # p beyond the 4 parameter tiles of one launch: the class surface solves it in column blocks (any p, like the reference, PDP.py:446-555)
    p2 = 150
    E2, Hxe2, Hue2, hxe2 = 0.1 * rng.standard_normal((B, T, n, p2)), 0.2 * rng.standard_normal((B, T, n, p2)), 0.2 * rng.standard_normal((B, T, m, p2)), 0.2 * rng.standard_normal((B, n, p2))
    X0 = rng.standard_normal((B, n, p2))
    X2, U2, L2, st2 = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe2, E=E2, Hxe=Hxe2, Hue=Hue2, X0=X0)
    assert int(st2.sum()) == 0 and X2.shape == (B, T + 1, n, p2) and U2.shape == (B, T, m, p2)
    sol2 = po.lqr_solver(list(F[0]), list(G[0]), list(E2[0]), list(Hxx[0]), list(Huu[0]), Z, list(Hxe2[0]), list(Hue2[0]), [hxx[0]], [hxe2[0]], X0[0], T)
    assert rel(npy(X2)[0], np.stack(sol2["state_traj_opt"])) < TOL and rel(npy(U2)[0], np.stack(sol2["control_traj_opt"])) < TOL
    assert rel(npy(L2)[0], np.stack(sol2["costate_traj_opt"])) < TOL
    # the C entry point itself reports what one launch cannot hold
    core = rt.load_core()
    pr = rt.PdpLqrProblem()
    pr.B, pr.T, pr.n, pr.m, pr.p = 1, 2, 4, 1, 64
    assert core.pdp_lqr_solve_batched(__import__("ctypes").byref(pr), None, None, None, None, None, 0, None) in (-1, -2)
    # since round 5 no state or control dimension is refused (n = 33, m = 9 used to be PDP_E_SIZE; parity at such sizes: tests/test_gpu_generic_sizes.py)
    X3, U3, _, st3 = rt.lqr_solve(np.tile(np.eye(33), (1, 2, 1, 1)), np.zeros((1, 2, 33, 1)), np.tile(np.eye(33), (1, 2, 1, 1)), np.ones((1, 2, 1, 1)), np.eye(33)[None], np.zeros((1, 33, 1)))
    assert int(st3.sum()) == 0 and X3.shape == (1, 3, 33, 1) and float(X3.abs().max()) == 0.0
    X4, U4, _, st4 = rt.lqr_solve(np.tile(np.eye(12), (1, 2, 1, 1)), np.zeros((1, 2, 12, 9)), np.tile(np.eye(12), (1, 2, 1, 1)), np.tile(np.eye(9), (1, 2, 1, 1)), np.eye(12)[None], np.zeros((1, 12, 1)))
    assert int(st4.sum()) == 0 and U4.shape == (1, 2, 9, 1)


@pytest.mark.parametrize("n,m,p", [(20, 3, 11), (17, 1, 5), (12, 6, 45), (32, 8, 32), (9, 5, 70)])
def test_lqr_beyond_one_tile_per_matrix(n, m, p):
    """16 < n <= 32 or 4 < m <= 8: the generic LDS kernel (lqr_solve_generic_kernel), any p through column blocks - the reference's
    lqrSolver accepts any size (PDP.py:446-555).  Against the numpy restatement, incl. costates, broadcast and optional inputs."""
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(7 * n + m)
    T, B = 7, 3

    def spd(k, s):
        A = rng.standard_normal((k, k))
        return s * (A @ A.T / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n)); G = 0.3 * rng.standard_normal((B, T, n, m)); E = 0.1 * rng.standard_normal((B, T, n, p))
    Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)]); Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
    Hxu = 0.05 * rng.standard_normal((B, T, n, m)); Hxe, Hue = 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
    hxx = np.stack([spd(n, 1.0) for _ in range(B)]); hxe, X0 = 0.2 * rng.standard_normal((B, n, p)), rng.standard_normal((B, n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0)
    assert int(st.sum()) == 0
    for b in range(B):
        sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]), [hxx[b]], [hxe[b]], X0[b], T)
        assert rel(npy(X)[b], np.stack(sol["state_traj_opt"])) < TOL and rel(npy(U)[b], np.stack(sol["control_traj_opt"])) < TOL
        assert rel(npy(Lam)[b], np.stack(sol["costate_traj_opt"])) < TOL
    X3, U3, _, st3 = rt.lqr_solve(F[0, 0], G[0, 0], Hxx[0, 0], Huu[0, 0], hxx, hxe, T=T, want_costate=False)
    Z = lambda r, c: T * [np.zeros((r, c))]
    sol = po.lqr_solver(T * [F[0, 0]], T * [G[0, 0]], Z(n, p), T * [Hxx[0, 0]], T * [Huu[0, 0]], Z(n, m), Z(n, p), Z(m, p), [hxx[1]], [hxe[1]], np.zeros((n, p)), T)
    assert int(st3.sum()) == 0 and rel(npy(X3)[1], np.stack(sol["state_traj_opt"])) < TOL and rel(npy(U3)[1], np.stack(sol["control_traj_opt"])) < TOL


def test_sysid_ragged_batch_and_empty_gradient_directions(golden_dir):
    """trajectories of different horizons in one SysID.step call (the reference loops over them one by one, PDP.py:1266-1291)"""
    from oracle import models, pdp_oracle as po
    from pdp_amd import PDP, zoo
    env, dt = zoo.make_env("cartpole", "sysid")
    sid = PDP.SysID()
    sid.setAuxvarVariable(env.dyn_auxvar); sid.setStateVariable(env.X); sid.setControlVariable(env.U); sid.setDyn(env.X + dt * env.f)
    st = models.SYSID_SETUP["cartpole"]
    ora = po.make_sysid(models.REGISTRY["cartpole"](**st["kwargs"]), st["dt"])
    rng = np.random.default_rng(3)
    inputs = [rng.uniform(-1, 1, (T, 1)) for T in (5, 12, 12, 1, 30)]
    states = [ora.integrateDyn([0, 0.1, 0, 0], u, [1.0, 1.0, 1.0]) for u in inputs]
    theta = np.array([1.2, 0.9, 1.1])
    loss, d = sid.step(inputs, states, theta)
    lo, do = ora.step(inputs, states, theta)
    assert abs(loss - lo) < 1e-11 * lo and rel(d, do) < TOL


def test_results_do_not_depend_on_stale_device_memory():
    """every kernel must read only what it was given / what it wrote: the same calls are repeated after the caching allocator's
    free blocks have been filled with NaNs and scattered differently (an out-of-bounds or uninitialised read then shows up as a NaN,
    a different number or a memory fault instead of passing by the luck of a fresh process)."""
    import torch
    from pdp_amd import runtime as rt, zoo
    import bench
    rng = np.random.default_rng(11)

    def lqr_case(n, m, p, T, B):
        r = np.random.default_rng(1000 * n + p)
        spd = lambda k, s: (lambda A: s * (A @ A.T / k + 0.5 * np.eye(k)))(r.standard_normal((k, k)))
        a = dict(F=np.eye(n) + 0.1 * r.standard_normal((B, T, n, n)), G=0.3 * r.standard_normal((B, T, n, m)),
                 Hxx=np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)]),
                 Huu=np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)]), hxx=np.stack([spd(n, 1.0) for _ in range(B)]),
                 hxe=0.2 * r.standard_normal((B, n, p)))
        k = dict(E=0.1 * r.standard_normal((B, T, n, p)), Hxu=0.05 * r.standard_normal((B, T, n, m)), Hxe=0.2 * r.standard_normal((B, T, n, p)),
                 Hue=0.2 * r.standard_normal((B, T, m, p)), X0=r.standard_normal((B, n, p)))
        return lambda: [t.cpu().numpy() for t in rt.lqr_solve(a["F"], a["G"], a["Hxx"], a["Huu"], a["hxx"], a["hxe"], **k)[:3]]

    mdl = zoo.get("quadrotor", "irl")
    x0, u, dx, du = bench.synth_inputs(48, 3)
    cp = zoo.get("quadrotor", "oc")
    pol = rt.make_policy("mlp", layers=[13, 13, 4])
    thm = 0.1 * rng.standard_normal(420)
    ocm = zoo.get("cartpole", "irl")
    xs0 = np.stack([np.zeros(24), rng.uniform(-0.4, 0.4, 24), np.zeros(24), np.zeros(24)], axis=1)
    ths = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0]) * (1 + 0.05 * rng.uniform(-1, 1, (24, 7)))

    def oc_solve():
        sol = ocm.oc_solve(xs0, torch.zeros((24, 20, 1), dtype=torch.float64, device="cuda"), ths, max_iter=60, want_gains=True)
        return [npy(sol[k]) for k in ("state", "control", "costate", "cost", "gains")] + [np.array(sol["iterations"])]

    calls = [lqr_case(13, 4, 9, 50, 64), lqr_case(7, 2, 40, 11, 5), lqr_case(4, 1, 1, 30, 40), oc_solve,
             lambda: [npy(v) for v in (lambda o: (o["loss"], o["grad"], o["x"], o["lam"]))(mdl.oc_pdp_grad(u, np.array(bench.THETA), dx, du, x0=x0))],
             lambda: [npy(v) for v in cp.cp_step(pol, 420, x0[:, :13], thm, 40)]]
    fresh = [c() for c in calls]
    junk = [torch.full((int(s),), float("nan"), dtype=torch.float64, device="cuda") for s in (3e5, 1e6, 2e6, 5e6, 3e6, 7e5, 4e6, 1e5)]
    del junk
    for c, ref in zip(calls, fresh):
        again = c()
        for a, b in zip(again, ref):
            assert np.array_equal(a, b)


def test_fused_unit_beyond_one_parameter_tile_takes_the_materialised_kernel_route():
    """m + p > 16 (here m = 2, p = 16): the fused kernel holds the control and parameter columns in ONE tile and returns PDP_E_SIZE; the
    class surface then runs the reference's own route kernel by kernel on the GPU (getAuxSys -> lqrSolver -> chain rule).  Checked against
    the numpy oracle on the kernels' aux matrices."""
    from oracle import pdp_oracle as po
    from pdp_amd import PDP
    from pdp_amd.sx import SX, mtimes, dot
    rng = np.random.default_rng(12)
    n, m, dt, T, B = 6, 2, 0.1, 9, 3
    A, Bm = rng.standard_normal((n, n)) - np.eye(n), rng.standard_normal((n, m))
    X, U, w = SX.sym("x", n), SX.sym("u", m), SX.sym("w", 16)
    f = X + dt * (mtimes(SX(A), X) + mtimes(SX(Bm), U) + w[8:14] * X * X)              # 6 dynamics parameters
    cost = sum(w[i] * X[i] * X[i] for i in range(n)) + w[6] * U[0] * U[0] + w[7] * U[1] * U[1] + w[14] * X[0] * U[0] + w[15] * X[1] * U[1]
    oc = PDP.OCSys("wide auxvar")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(X)
    oc.setControlVariable(U)
    oc.setDyn(f)
    oc.setPathCost(cost)
    oc.setFinalCost(sum(w[i] * X[i] * X[i] for i in range(n)))
    th = np.concatenate([1 + rng.random(8), 0.05 * rng.standard_normal(6), 0.1 * rng.standard_normal(2)])
    x0, u = 0.5 * rng.standard_normal((B, n)), 0.3 * rng.standard_normal((B, T, m))
    xs, _ = oc.rollout_batch(x0, u, th)
    demo_x, demo_u = npy(xs) + 0.1 * rng.standard_normal(xs.shape), u + 0.1 * rng.standard_normal(u.shape)
    out = oc.pdp_grad_batch(u, th, demo_x, demo_u, ini_state=x0, want_sens=True)
    assert int(out["status"].sum()) == 0 and out["grad"].shape == (B, 16)
    aux = oc.getAuxSys_batch(out["x"], u, out["lam"], th)
    for i in range(B):
        a = {k: [np.asarray(v) for v in npy(aux[k])[i]] for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue")}
        a["hxx"], a["hxe"] = [npy(aux["hxx"])[i]], [npy(aux["hxe"])[i]]
        ref = po.lqr_from_aux(a, n, 16, T)
        assert rel(npy(out["dxdp"])[i], np.stack(ref["state_traj_opt"])) < TOL and rel(npy(out["dudp"])[i], np.stack(ref["control_traj_opt"])) < TOL
        l, g = po.irl_loss_grad(npy(out["x"])[i], u[i], demo_x[i], demo_u[i], ref["state_traj_opt"], ref["control_traj_opt"])
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < TOL
    # and the gradient is the derivative of the loss through the trajectory only where theta enters the rollout: finite difference on a
    # dynamics parameter with the controls fixed differs (the PDP gradient differentiates the OPTIMAL control problem), so only shapes / finiteness here
    assert np.all(np.isfinite(npy(out["grad"])))


def test_long_horizon_fused_unit_equals_the_materialised_route():
    """T = 600 (the fused kernel's rollout staging: 82 KB of LDS per wave, 24 chunks) and T = 1500 (staging beyond the 160 KB of a CU: the
    class surface takes the kernel-by-kernel route): both equal getAuxSys -> lqrSolver -> chain rule on the same trajectories."""
    from pdp_amd import zoo
    mdl = zoo.get("quadrotor", "irl")
    rng = np.random.default_rng(9)
    th = np.array([1, 1, 1, 1, .4, 1, 1, 5, 1.0])
    for T in (600, 1500):
        B = 2
        x0 = np.zeros((B, 13)); x0[:, 2] = 1.0; x0[:, 6] = 1.0
        u = np.full((B, T, 4), 2.5)                  # hover thrust: the open-loop trajectory stays put over 60 / 150 s (a perturbed one leaves
        u[1, :, 0] += 1e-7                           # the region where the LQ problem along it is well conditioned, and 1e-9 costate differences
                                                     # between the two routes would be amplified to 1e-3 in the gradient)
        dx = np.zeros((B, T + 1, 13)); dx[:, :, 6] = 1.0
        du = np.full((B, T, 4), 2.5)
        a = mdl.oc_pdp_grad(u, th, dx, du, x0=x0)
        b = mdl.oc_pdp_grad_materialised(u, th, dx, du, x0=x0)
        assert int(a["status"].sum()) == 0 and int(b["status"].sum()) == 0
        assert rel(npy(a["x"]), npy(b["x"])) < 1e-9 and rel(npy(a["lam"]), npy(b["lam"])) < 1e-9
        assert rel(npy(a["loss"]), npy(b["loss"])) < 1e-12 and rel(npy(a["grad"]), npy(b["grad"])) < 1e-7


def test_user_model_with_20_states_runs_the_whole_class_surface():
    """n = 20, m = 5, p = 6 (a chain of ten masses with five actuators): beyond the tile kernels (n <= 16, m <= 4).  rollout, costates and
    getAuxSys are size-generic, lqrSolver takes the generic LDS kernel, pdp_grad_batch the kernel-by-kernel route: against the numpy oracle."""
    from oracle import pdp_oracle as po
    from pdp_amd import PDP
    from pdp_amd.sx import SX, vertcat, dot
    rng = np.random.default_rng(21)
    nm, m, dt, T, B = 10, 5, 0.05, 8, 3
    q, v, U, w = SX.sym("q", nm), SX.sym("v", nm), SX.sym("u", m), SX.sym("w", 6)       # w: stiffness, damping, 4 cost weights
    acc = []
    for i in range(nm):
        left = q[i - 1] if i > 0 else 0.0
        right = q[i + 1] if i + 1 < nm else 0.0
        a = w[0] * (left - 2 * q[i] + right) - w[1] * v[i] - 0.3 * q[i] * q[i] * q[i]
        if i % 2 == 0:
            a = a + U[i // 2]
        acc.append(a)
    X = vertcat(q, v)
    f = X + dt * vertcat(v, vertcat(*acc))
    oc = PDP.OCSys("mass chain")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(X)
    oc.setControlVariable(U)
    oc.setDyn(f)
    oc.setPathCost(w[2] * dot(q, q) + w[3] * dot(v, v) + w[4] * dot(U, U))
    oc.setFinalCost(w[5] * dot(q, q))
    th = np.array([2.0, 0.3, 1.0, 0.5, 0.2, 3.0])
    x0, u = 0.5 * rng.standard_normal((B, 2 * nm)), 0.3 * rng.standard_normal((B, T, m))
    xs, _ = oc.rollout_batch(x0, u, th)
    demo_x, demo_u = npy(xs) + 0.1 * rng.standard_normal(xs.shape), u + 0.1 * rng.standard_normal(u.shape)
    out = oc.pdp_grad_batch(u, th, demo_x, demo_u, ini_state=x0, want_sens=True)
    assert int(out["status"].sum()) == 0 and out["grad"].shape == (B, 6)
    aux = oc.getAuxSys_batch(out["x"], u, out["lam"], th)
    n = 2 * nm
    for i in range(B):
        a = {k: [np.asarray(mm) for mm in npy(aux[k])[i]] for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue")}
        a["hxx"], a["hxe"] = [npy(aux["hxx"])[i]], [npy(aux["hxe"])[i]]
        ref = po.lqr_from_aux(a, n, 6, T)
        assert rel(npy(out["dxdp"])[i], np.stack(ref["state_traj_opt"])) < TOL and rel(npy(out["dudp"])[i], np.stack(ref["control_traj_opt"])) < TOL
        l, g = po.irl_loss_grad(npy(out["x"])[i], u[i], demo_x[i], demo_u[i], ref["state_traj_opt"], ref["control_traj_opt"])
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < TOL
    # ocSolver for the same model (single shooting, the LQ step on the generic kernel): a KKT point - H_u = 0 along rollout and costates
    sol = oc.ocSolver_batch(x0, T, th)
    assert bool(sol["converged"].all())
    hu = oc.model().oc_auxsys(sol["state"], sol["control"], sol["costate"], th, only=("dHu",))["dHu"]
    assert float(hu.abs().max()) <= 1e-8 * (1 + float(sol["control"].abs().max()))
    assert float((oc.rollout_batch(x0, sol["control"], th)[0] - sol["state"]).abs().max()) <= 1e-10
    # the aux matrices themselves: finite differences of the kernels' own rollout (F) on one sample
    eps = 1e-6
    xp, xm = x0[:1].copy(), x0[:1].copy()
    xp[0, 3] += eps
    xm[0, 3] -= eps
    fd = (npy(oc.rollout_batch(xp, u[:1, :1], th)[0])[0, 1] - npy(oc.rollout_batch(xm, u[:1, :1], th)[0])[0, 1]) / (2 * eps)
    F0 = npy(oc.getAuxSys_batch(npy(xs)[:1, :2], u[:1, :1], np.zeros((1, 1, n)), th)["dynF"])[0, 0]
    assert np.abs(F0[:, 3] - fd).max() < 1e-8


def _mass_chain(nm, m, dt, lib):
    """the mass chain of the test above in a symbolic library `lib` with the calls both have (pdp_amd.sx for the product, sympy for the oracle)"""
    if lib == "sx":
        from pdp_amd.sx import SX, vertcat, dot
        q, v, U, w = SX.sym("q", nm), SX.sym("v", nm), SX.sym("u", m), SX.sym("w", 6)
        qs, vs, us, ws = [q[i] for i in range(nm)], [v[i] for i in range(nm)], [U[i] for i in range(m)], [w[i] for i in range(6)]
    else:
        import sympy as sp
        qs, vs, us, ws = (list(sp.symbols("%s0:%d" % (nme, k), real=True)) for nme, k in (("q", nm), ("v", nm), ("u", m), ("w", 6)))
    acc = []
    for i in range(nm):
        left = qs[i - 1] if i > 0 else 0.0
        right = qs[i + 1] if i + 1 < nm else 0.0
        a = ws[0] * (left - 2 * qs[i] + right) - ws[1] * vs[i] - 0.3 * qs[i] * qs[i] * qs[i]
        if i % 2 == 0:
            a = a + us[i // 2]
        acc.append(a)
    sq = lambda xs: sum((z * z for z in xs[1:]), xs[0] * xs[0])
    path, final = ws[2] * sq(qs) + ws[3] * sq(vs) + ws[4] * sq(us), ws[5] * sq(qs)
    f = [qs[i] + dt * vs[i] for i in range(nm)] + [vs[i] + dt * acc[i] for i in range(nm)]
    return qs + vs, us, ws, f, path, final


def test_multiple_shooting_route_for_20_states_follows_the_oracle():
    """ocSolver for n = 20, m = 5 (beyond the solver kernel's tiles) runs the SAME multiple-shooting iteration kernel by kernel - NLP residuals
    (pdp_oc_ms_residuals_batched), KKT matrices (pdp_oc_auxsys_batched), Newton step on the generic LQR kernel with its positive-definiteness report
    (PDP_STATUS_INDEFINITE), IPOPT's bookkeeping on tensors (ocsolver.solve_batch_ms_generic) - and follows the CPU restatement of IPOPT's algorithm
    (oracle/ipopt_ms.py) iteration by iteration: same inertia corrections, same step lengths, same solution.  Before round 3 these sizes were handed to single shooting."""
    import sympy as sp
    from oracle import ipopt_ms, pdp_oracle as po
    from pdp_amd import PDP, ocsolver
    from pdp_amd.sx import vertcat
    nm, m, dt, B = 10, 5, 0.05, 3
    X, U, w, f, path, final = _mass_chain(nm, m, dt, "sx")
    oc = PDP.OCSys("mass chain ms")
    oc.setAuxvarVariable(vertcat(*w))
    oc.setStateVariable(vertcat(*X))
    oc.setControlVariable(vertcat(*U))
    oc.setDyn(vertcat(*f))
    oc.setPathCost(path)
    oc.setFinalCost(final)
    Xs, Us, ws_, fs, paths, finals = _mass_chain(nm, m, dt, "sympy")
    ref_oc = po.OCSysOracle(sp.Matrix(Xs), sp.Matrix(Us), list(ws_), sp.Matrix(fs), paths, finals)
    # (a) convex weights, far initial states: step lengths below 1 in the first iterations; (b) a negative state weight: inertia corrections on most iterations
    seen_dw, seen_alpha = False, False
    for scale, T, th in ((5.0, 30, np.array([2.0, 0.3, 1.0, 0.5, 0.2, 3.0])), (2.0, 25, np.array([2.0, 0.3, -2.0, 0.5, 0.02, 3.0]))):
        rng = np.random.default_rng(5)
        x0 = (scale * rng.standard_normal((B, 2 * nm)))[:2 if scale == 5.0 else B]
        out = ocsolver.solve_batch_ms_generic(oc, x0, T, th, tol=1e-10, log_rows=40)
        assert bool(out["converged"].all()) and int((out["status"] & ~1024).sum()) == 0            # (1024: PDP_MS_SOC, informational)
        for b in range(x0.shape[0]):
            log = []
            ref = ipopt_ms.solve(ref_oc, x0[b], T, th, tol=1e-10, log=log)
            assert int(out["iterations"][b]) == ref["iterations"] == len(log)
            kl = npy(out["log"])[b]
            for r_, l in zip(kl, log):
                assert r_[5] == (-l["alpha"] if l["soc_taken"] else l["alpha"]) and abs(r_[4] - l["dw"]) <= 1e-12 * max(1.0, l["dw"]), (b, l["it"], r_[4], r_[5], l["dw"], l["alpha"])
                assert abs(r_[1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(r_[2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
                seen_dw, seen_alpha = seen_dw or l["dw"] > 0.0, seen_alpha or l["alpha"] < 1.0
            for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
                assert np.abs(npy(out[k])[b] - ref[kr]).max() <= 1e-8 * max(1.0, np.abs(ref[kr]).max())
    assert seen_dw and seen_alpha
    # the second-order correction on this route (soc=True, off by default like PDP_MS_WITH_SOC in the solver kernel): per-sample masks - the two far initial states of (a)
    # try 7 and 4 corrections and take one each, at different iterations - row by row the restatement's with the same switch; 9 and 8 iterations instead of 13 and 9
    scale, T, th = 5.0, 30, np.array([2.0, 0.3, 1.0, 0.5, 0.2, 3.0])
    x0 = (scale * np.random.default_rng(5).standard_normal((B, 2 * nm)))[:2]
    out = ocsolver.solve_batch_ms_generic(oc, x0, T, th, tol=1e-10, log_rows=40, soc=True)
    assert bool(out["converged"].all()) and (npy(out["status"]) == 1024).all() and (npy(out["iterations"]) == [9, 8]).all()
    for b in range(2):
        log = []
        ref = ipopt_ms.solve(ref_oc, x0[b], T, th, tol=1e-10, log=log, soc=True)
        assert int(out["iterations"][b]) == ref["iterations"] == len(log) and ref["soc_steps"] == (7, 4)[b] and sum(1 for l in log if l["soc_taken"]) == 1
        for r_, l in zip(npy(out["log"])[b], log):
            assert r_[5] == (-l["alpha"] if l["soc_taken"] else l["alpha"]) and abs(r_[4] - l["dw"]) <= 1e-12 * max(1.0, l["dw"]), (b, l["it"], r_[4], r_[5], l["dw"], l["alpha"])
            assert abs(r_[1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(r_[2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
        for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
            assert np.abs(npy(out[k])[b] - ref[kr]).max() <= 1e-8 * max(1.0, np.abs(ref[kr]).max())
    # the class surface takes this route for such sizes (and reports it as the multiple-shooting method)
    sol = oc.ocSolver_batch(x0, T, th)
    assert bool(sol["converged"].all()) and bool(sol["method_ms"].all())
    assert float((sol["state"] - out["state"]).abs().max()) <= 1e-8


def test_oc_solver_small_state_many_controls_takes_the_generic_lq_kernel():
    """n = 3 <= 4 but m = 5 > 4: the packed small-system LQ kernel keeps the m x m control block in four tile rows, so this shape must reach the
    generic LDS kernel (round-2 advisor finding: the dispatch looked at n alone and silently produced wrong Newton steps).  The problem is
    linear-quadratic, so the optimum is the solution of one KKT system: compared against numpy."""
    from pdp_amd import PDP
    from pdp_amd.sx import SX, dot, mtimes
    rng = np.random.default_rng(33)
    n, m, T, B = 3, 5, 12, 4
    A = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    Bm = 0.5 * rng.standard_normal((n, m))
    X, U, w = SX.sym("x", n), SX.sym("u", m), SX.sym("w", 2)
    oc = PDP.OCSys("wide")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(X)
    oc.setControlVariable(U)
    oc.setDyn(mtimes(SX(A), X) + mtimes(SX(Bm), U))
    oc.setPathCost(w[0] * dot(X, X) + w[1] * dot(U, U))
    oc.setFinalCost(3.0 * w[0] * dot(X, X))
    th = np.array([1.3, 0.4])
    x0 = rng.standard_normal((B, n))
    sol = oc.ocSolver_batch(x0, T, th, method="single")
    assert bool(sol["converged"].all())
    # numpy: minimise sum q |x_t|^2 + r |u_t|^2 + 3 q |x_T|^2 subject to the dynamics, by condensing onto the controls
    q, r = th
    for i in range(B):
        Phi = np.zeros(((T + 1) * n, T * m))
        free = np.zeros(((T + 1) * n,))
        xk = x0[i].copy()
        free[:n] = xk
        for t in range(T):
            xk = A @ xk
            free[(t + 1) * n:(t + 2) * n] = xk
        for s in range(T):
            blk = Bm.copy()
            for t in range(s + 1, T + 1):
                Phi[t * n:(t + 1) * n, s * m:(s + 1) * m] = blk
                blk = A @ blk
        Wx = np.full((T + 1) * n, q)
        Wx[T * n:] = 3.0 * q
        Hm = Phi.T @ (Wx[:, None] * Phi) + r * np.eye(T * m)
        u_star = np.linalg.solve(Hm, -Phi.T @ (Wx * free)).reshape(T, m)
        x_star = (free + Phi @ u_star.reshape(-1)).reshape(T + 1, n)
        assert np.abs(npy(sol["control"])[i] - u_star).max() <= 1e-9 * (1 + np.abs(u_star).max())
        assert np.abs(npy(sol["state"])[i] - x_star).max() <= 1e-9 * (1 + np.abs(x_star).max())


@pytest.mark.parametrize("T", [1, 7, 70])
def test_large_batch_routes_with_ragged_sizes_equal_the_small_batch_routes(T):
    """the rollout pre-pass of SysID.step / ControlPlanning.step (batches beyond two trajectories per SIMD: one lane per trajectory, then the given-trajectory kernels) on a
    batch that is no multiple of anything, at horizons from a single step to more than a wavefront's lanes: the same numbers as the same trajectories sent in
    sub-batches that take the in-kernel rollouts; the MLP step's two-wavefronts-per-SIMD layout likewise"""
    from pdp_amd import JinEnv, runtime as rt, zoo
    rng = np.random.default_rng(T)
    B = 2048 + 131
    cuts = [(0, 1000), (1000, 2048), (2048, B)]
    # SysID.step
    sid = zoo.get("cartpole", "sysid")
    assert int(sid.lib.pdp_sysid_step_workspace_bytes(B, T)) > 0 and int(sid.lib.pdp_sysid_step_workspace_bytes(2048, T)) == 0
    u = rng.uniform(-1, 1, (B, T, 1))
    x0 = np.tile(np.array([0, 0.1, 0, 0.0]), (B, 1)) + 0.05 * rng.standard_normal((B, 4))
    xobs = npy(sid.sysid_integrate(x0, u, np.array([1.0, 1.0, 1.0])))
    th = np.array([1.2, 0.9, 1.1])
    L, G = (npy(a) for a in sid.sysid_step(u, xobs, th))
    for lo, hi in cuts:
        l2, g2 = (npy(a) for a in sid.sysid_step(u[lo:hi], xobs[lo:hi], th))
        assert np.array_equal(g2, G[lo:hi]) and np.abs(l2 - L[lo:hi]).max() <= 1e-14 * max(1e-300, np.abs(L).max())
    # ControlPlanning.step, Lagrange policy (per-sample parameters)
    cp = zoo.get("pendulum", "oc")
    npiv = 4 if T >= 3 else 2
    pol = rt.make_policy("poly", pivots=np.linspace(0, T, npiv))
    p = npiv * cp.m
    thp = 0.3 * rng.standard_normal((B, p))
    xc0 = rng.uniform(-0.5, 0.5, (B, cp.n))
    assert int(cp.lib.pdp_cp_step_workspace_bytes(B, T, __import__("ctypes").byref(pol), p)) == B * ((T + 1) * cp.n + T * cp.m + cp.n) * 8
    Lc, Gc, Xc, Uc = (npy(a) for a in cp.cp_step(pol, p, xc0, thp, T, want_traj=True))
    L0, G0 = (npy(a) for a in cp.cp_step(pol, p, xc0, thp, T))
    assert np.array_equal(L0, Lc) and np.array_equal(G0, Gc)
    for lo, hi in cuts:
        l2, g2, x2, u2 = (npy(a) for a in cp.cp_step(pol, p, xc0[lo:hi], thp[lo:hi], T, want_traj=True))
        assert np.array_equal(l2, Lc[lo:hi]) and np.array_equal(g2, Gc[lo:hi]) and np.array_equal(x2, Xc[lo:hi]) and np.array_equal(u2, Uc[lo:hi])
    # ControlPlanning.step, tanh MLP (shared parameters): 20 KB layout above one trajectory per SIMD, 40 KB below
    if T > 1:
        qd = zoo.get("cartpole", "oc")
        polm = rt.make_policy("mlp", layers=[qd.n, qd.n, qd.m])
        pm = 2 * (qd.n * qd.n + qd.n) + qd.m * qd.n + qd.m          # three weight layers: n x n, n x n, m x n (+ biases)
        thm = 0.1 * rng.standard_normal(pm)
        xq = 0.2 * rng.standard_normal((B, qd.n))
        Lm, Gm = (npy(a) for a in qd.cp_step(polm, pm, xq, thm, T))
        for lo, hi in cuts[:2]:
            l2, g2 = (npy(a) for a in qd.cp_step(polm, pm, xq[lo:hi], thm, T))
            assert np.array_equal(l2, Lm[lo:hi]) and np.array_equal(g2, Gm[lo:hi])
