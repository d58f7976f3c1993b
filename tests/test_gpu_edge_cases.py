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
    """n = 16, m = 4, p = 60 (= 64 - m) is the largest problem of pdp_lqr_solve_batched; beyond it the C entry point returns PDP_E_SIZE"""
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
    with pytest.raises(RuntimeError, match="PDP_E_SIZE"):
        rt.lqr_solve(F, G, Hxx, Huu, hxx, np.zeros((B, n, 61)), E=np.zeros((B, T, n, 61)))
    with pytest.raises(RuntimeError, match="PDP_E_SIZE"):
        rt.lqr_solve(np.zeros((1, 2, 17, 17)), np.zeros((1, 2, 17, 1)), np.zeros((1, 2, 17, 17)), np.ones((1, 2, 1, 1)), np.zeros((1, 17, 17)), np.zeros((1, 17, 1)))


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
