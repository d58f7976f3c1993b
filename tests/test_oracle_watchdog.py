"""CPU: the EXPERIMENTAL watchdog of the IPOPT restatement (oracle/ipopt_ms.py: solve(watchdog=True); the checker of PDP_MS_WITH_WATCHDOG, tests/test_gpu_ms_watchdog.py).
IPOPT has a watchdog on by default; this restatement of it is unpinned (no IPOPT here), so what can be tested on the CPU is its own logic: it shortens a solve that crawls,
ends in the same optimum, and is silent on every stored demonstration of the reference (their solves never take ten shortened steps in a row)."""
import os

import numpy as np

from oracle import ipopt_ms, models, pdp_oracle as po


def _oc(name):
    st = models.IRL_SETUP[name]
    return po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])


def test_watchdog_shortens_a_crawling_cold_solve_and_keeps_its_optimum():
    rng = np.random.default_rng(0)
    rng.uniform(-0.5, 0.5, 256); rng.uniform(-0.45, 0.45, (256, 7))          # (problem 15 of probes/solver_robustness.py's C4 case)
    x0 = np.zeros((512, 13))
    x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((512, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = models.to_quaternion(1.5, [0, 0, 1])
    th = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    oc = _oc("rocket")
    la, lw = [], []
    a = ipopt_ms.solve(oc, x0[15], 100, th, tol=1e-8, max_iter=300, log=la)
    w = ipopt_ms.solve(oc, x0[15], 100, th, tol=1e-8, max_iter=300, log=lw, watchdog=True)
    assert a["iterations"] == 116 and w["iterations"] == 65 and a["watchdog_starts"] == 0 and w["watchdog_starts"] >= 1
    assert abs(a["cost"] - w["cost"]) <= 1e-6 * abs(a["cost"])
    # identical up to the first procedure; the trigger: ten accepted steps with alpha < 1 in a row right in front of it
    first = next(i for i, l in enumerate(lw) if l.get("wd"))
    assert all(lw[i]["alpha"] == la[i]["alpha"] and lw[i]["f"] == la[i]["f"] for i in range(first))
    assert all(0.0 < lw[i]["alpha"] < 1.0 for i in range(first - 10, first)) and lw[first]["alpha"] == 1.0
    tags = [l.get("wd") for l in lw]
    assert "success" in tags or "stop" in tags


def test_watchdog_is_silent_on_the_stored_demonstrations(golden_dir):
    for name in ("pendulum", "cartpole", "robotarm", "quadrotor", "rocket"):
        d = np.load(os.path.join(golden_dir, "demos_%s.npz" % name))
        oc = _oc(name)
        T = d["control"].shape[1]
        a = ipopt_ms.solve(oc, d["state"][0, 0], T, d["true_parameter"])
        w = ipopt_ms.solve(oc, d["state"][0, 0], T, d["true_parameter"], watchdog=True)
        assert w["watchdog_starts"] == 0 and w["iterations"] == a["iterations"], name
        assert np.array_equal(w["control_traj_opt"], a["control_traj_opt"]), name
