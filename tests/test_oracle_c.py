"""CPU: the C restatement (oracle/pdp_oracle.c, timed as bench.py's cpu_baseline) agrees with the numpy oracle, which is
pinned against the reference in test_oracle_golden.py - so the timed CPU baseline computes the reference's results."""
import os

import numpy as np
import pytest

from oracle import c_oracle, models, pdp_oracle as po

SYSTEMS = ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"]


@pytest.fixture(scope="module")
def lib():
    return c_oracle.load()


@pytest.mark.parametrize("name", SYSTEMS)
def test_c_oracle_matches_numpy_oracle_on_demos(golden_dir, lib, name):
    d = np.load(os.path.join(golden_dir, "demos_%s.npz" % name))
    rl = np.load(os.path.join(golden_dir, "ref_lqr_%s.npz" % name))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    rng = np.random.default_rng(1)
    demo_x = d["state"] + 0.1 * rng.standard_normal(d["state"].shape)
    demo_u = d["control"] + 0.1 * rng.standard_normal(d["control"].shape)
    # given optimal trajectory: sensitivities equal the reference's own lqrSolver output
    out = c_oracle.oc_unit(lib, name, d["control"], d["true_parameter"], demo_x, demo_u, x=d["state"], lam=d["costate"], want_sens=True)
    assert np.abs(out["dxdp"] - rl["X"]).max() <= 1e-9 * max(1, np.abs(rl["X"]).max())
    assert np.abs(out["dudp"] - rl["U"]).max() <= 1e-9 * max(1, np.abs(rl["U"]).max())
    # rollout mode vs the numpy oracle's unit
    th = d["true_parameter"] * 1.03
    u = d["control"] * 0.99
    out = c_oracle.oc_unit(lib, name, u, th, d["state"], d["control"], x0=d["state"][:, 0], threads=2)
    for i in range(d["state"].shape[0]):
        o = po.pdp_oc_unit(oc, d["state"][i, 0], u[i], th, d["state"][i], d["control"][i])
        assert np.abs(out["x"][i] - o["state_traj"]).max() < 1e-9 * max(1, np.abs(o["state_traj"]).max())
        assert abs(out["loss"][i] - o["loss"]) < 1e-9 * abs(o["loss"])
        assert np.abs(out["grad"][i] - o["grad"]).max() < 1e-6 * max(1, np.abs(o["grad"]).max())   # fp64 reference order: see lqr_solver_mp
