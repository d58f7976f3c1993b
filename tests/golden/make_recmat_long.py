#!/usr/bin/env python3
"""Build-container-only: the reference's OWN ControlPlanning.recmat_init_step(horizon, -1) / recmat_step / recmat_unwarp (PDP/PDP.py:1039-1141,
imported unmodified from /root/reference) at the horizons its drivers use - Examples/OC/rocket/rocket_PDP_Recmat.py (T = 50),
Examples/OC/quadrotor/uav_PDP_Recmat.py (T = 35), Examples/OC/robotarm/robotarm_PDP_Recmat.py (T = 20) - stored as fixtures (data only).

The recovery matrix is ONE symbolic expression over the whole horizon; the sympy-backed CasADi stand-in of make_ref_outputs.py composes it up to T = 7
in about a minute and not at all at T = 20.  Round 3 generated these fixtures with the product's own expression engine (pdp_amd/sx.py) standing in for CasADi, which made
the AD engine common to fixture and product.  Since round 4 the `casadi` the reference imports here is casadi_numeric_shim.py: lazy matrix-valued graphs evaluated on
forward-mode dual numbers (values and Jacobians as numpy arrays) - no code, data structure or differentiation method in common with sx.py (scalar hash-consed DAG, reverse
mode).  The script (i) cross-checks that stand-in against the sympy-generated short fixtures (pendulum; rocket / quadrotor at T = 7), (ii) reports how far the
round-3 long fixtures (sx.py engine) are from the new numbers - three engines, one answer -, (iii) writes the long fixtures from the numeric stand-in.
`--engine sx` re-runs step (i) and the long cases with the product's engine instead (writes nothing)."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PDP_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    sys.exit("reference not present: fixtures can only be regenerated in the build container")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import matplotlib
matplotlib.use("Agg")
ENGINE = "sx" if "--engine" in sys.argv and sys.argv[sys.argv.index("--engine") + 1] == "sx" else "numeric"
if ENGINE == "sx":
    from pdp_amd import sx
    sx.numpy = np
    sx.np = np
    sx.casadi = sx
    sys.modules["casadi"] = sx
else:
    import casadi_numeric_shim
    casadi_numeric_shim.install()
sys.path.insert(0, REF)
from PDP import PDP                  # noqa: E402  (reference, unmodified)
from JinEnv import JinEnv            # noqa: E402  (reference, unmodified)


def make(name):
    if name == "rocket":             # Examples/OC/rocket/rocket_PDP_Recmat.py:10-28
        env = JinEnv.Rocket()
        env.initDyn(Jx=0.5, Jy=1, Jz=1, mass=1, l=1)
        env.initCost(wr=1, wv=1, wtilt=50, ww=1, wsidethrust=1, wthrust=0.4)
        x0 = [10.0, -8.0, 5.0, -0.1, 0.0, 0.0] + list(JinEnv.toQuaternion(1.5, [0, 0, 1])) + [0.0, 0.0, 0.0]
    elif name == "quadrotor":        # Examples/OC/quadrotor/uav_PDP_Recmat.py
        env = JinEnv.Quadrotor()
        env.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01)
        env.initCost(wr=1, wv=1, wq=5, ww=1, wthrust=0.1)
        x0 = [-8.0, -6.0, 9.0, 0.0, 0.0, 0.0] + list(JinEnv.toQuaternion(0, [1, -1, 1])) + [0.0, 0.0, 0.0]
    elif name == "robotarm":         # Examples/OC/robotarm/robotarm_PDP_Recmat.py:17-19
        env = JinEnv.RobotArm()
        env.initDyn(l1=1, m1=1, l2=1, m2=1, g=0)
        env.initCost(wq1=0.1, wq2=0.1, wdq1=0.1, wdq2=0.1, wu=0.01)
        x0 = [np.pi / 4, np.pi / 2, 0.0, 0.0]
    else:
        env = JinEnv.SinglePendulum()
        env.initDyn(l=1, m=1, damping_ratio=0.05)
        env.initCost(wq=10, wdq=1, wu=0.1)
        x0 = [0.0, 0.0]
    return env, np.array(x0, float)


def run(name, dt, T, grid, x0, theta=None, seed=0):
    env, x0_default = make(name)
    x0 = x0_default if x0 is None else x0
    cp = PDP.ControlPlanning()
    cp.setStateVariable(env.X)
    cp.setControlVariable(env.U)
    cp.setDyn(env.X + dt * env.f)
    cp.setPathCost(env.path_cost)
    cp.setFinalCost(env.final_cost)
    cp.recmat_init_step(T) if grid is None else cp.recmat_init_step(T, grid)
    if theta is None:
        theta = 0.5 * np.random.default_rng(seed).standard_normal(cp.n_auxvar)
    loss, grad = cp.recmat_step(x0, T, theta)
    un = cp.recmat_unwarp(x0, T, theta)
    return dict(dt=dt, T=T, x0=x0, theta=theta, grid=(-2 if grid is None else grid), time_grid=cp.time_grid, loss=float(np.asarray(loss).squeeze()),
                grad=np.asarray(grad, float).flatten(), state=un["state_traj"], control=un["control_traj"], cost=float(np.asarray(un["cost"]).squeeze()))


if __name__ == "__main__":
    print("engine standing in for CasADi:", ENGINE)
    # (i) cross-check of the stand-in against the sympy-generated fixtures
    for f, name in (("ref_recmat_pendulum_0.npz", "pendulum"), ("ref_recmat_rocket_2.npz", "rocket"), ("ref_recmat_quadrotor_3.npz", "quadrotor")):
        g = np.load(os.path.join(HERE, f))
        grid = None if int(g["grid"]) == -2 else int(g["grid"])
        r = run(name, float(g["dt"]), int(g["T"]), grid, g["x0"], g["theta"])
        el, eg = abs(r["loss"] - float(g["loss"])) / abs(float(g["loss"])), np.abs(r["grad"] - g["grad"]).max() / np.abs(g["grad"]).max()
        print("cross-check vs sympy engine  %-28s loss %.1e gradient %.1e (relative)" % (f, el, eg))
        assert el < 1e-12 and eg < 1e-12, "the stand-in disagrees with the sympy stand-in"
    for k, (name, T) in enumerate((("rocket", 50), ("quadrotor", 35), ("robotarm", 20))):
        r = run(name, 0.1, T, -1, None, seed=100 + k)
        path = os.path.join(HERE, "ref_recmat_long_%s.npz" % name)
        if os.path.exists(path):                             # (ii) what is stored (round 3: sx.py engine; later: this engine) against this run
            g = np.load(path)
            assert np.array_equal(g["theta"], r["theta"]) and np.array_equal(g["x0"], r["x0"])
            print("stored fixture vs this engine %-10s T %d: loss %.1e gradient %.1e state %.1e (relative)" %
                  (name, T, abs(r["loss"] - float(g["loss"])) / abs(r["loss"]), np.abs(r["grad"] - g["grad"]).max() / np.abs(r["grad"]).max(),
                   np.abs(r["state"] - g["state"]).max() / np.abs(r["state"]).max()))
        if ENGINE == "numeric":
            r["engine"] = np.array(2)                         # 2 = casadi_numeric_shim (forward-mode duals); the round-3 files (no such field) came from pdp_amd/sx.py
            np.savez_compressed(path, **r)
        print("recmat", name, "T", T, "p", r["theta"].size, "loss", r["loss"], "|grad|", np.abs(r["grad"]).max())
