#!/usr/bin/env python3
"""Build-container-only: the reference's OWN ControlPlanning.recmat_init_step(horizon, -1) / recmat_step / recmat_unwarp (PDP/PDP.py:1039-1141,
imported unmodified from /root/reference) at the horizons its drivers use - Examples/OC/rocket/rocket_PDP_Recmat.py (T = 50),
Examples/OC/quadrotor/uav_PDP_Recmat.py (T = 35), Examples/OC/robotarm/robotarm_PDP_Recmat.py (T = 20) - stored as fixtures (data only).

The recovery matrix is ONE symbolic expression over the whole horizon; the sympy-backed CasADi stand-in of make_ref_outputs.py composes it up to T = 7
in about a minute and not at all at T = 20.  Here the `casadi` the reference imports is this repository's SX-compatible expression DAG (pdp_amd/sx.py:
hash-consed nodes, reverse-mode AD - the same kind of object CasADi's SX is), which composes T = 50 in seconds.  That makes the AD engine common to the
fixture and the product, so the script first CROSS-CHECKS the stand-in: it re-runs the T = 7 rocket / quadrotor cases and the pendulum case that the sympy
stand-in generated (ref_recmat_*.npz) and requires agreement to 1e-12.  What the fixtures then pin is the reference's ALGORITHM (whole-horizon recovery
matrix) against the product's (one adjoint sweep on the GPU): different formulas for the same gradient."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PDP_REFERENCE", "/root/reference")
if not os.path.isdir(REF):
    sys.exit("reference not present: fixtures can only be regenerated in the build container")
sys.path.insert(0, ROOT)
import matplotlib
matplotlib.use("Agg")
from pdp_amd import sx
sx.numpy = np
sx.np = np
sx.casadi = sx
sys.modules["casadi"] = sx
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
    # cross-check of the stand-in against the sympy-generated fixtures
    for f, name in (("ref_recmat_pendulum_0.npz", "pendulum"), ("ref_recmat_rocket_2.npz", "rocket"), ("ref_recmat_quadrotor_3.npz", "quadrotor")):
        g = np.load(os.path.join(HERE, f))
        grid = None if int(g["grid"]) == -2 else int(g["grid"])
        r = run(name, float(g["dt"]), int(g["T"]), grid, g["x0"], g["theta"])
        el, eg = abs(r["loss"] - float(g["loss"])) / abs(float(g["loss"])), np.abs(r["grad"] - g["grad"]).max() / np.abs(g["grad"]).max()
        print("cross-check %-28s loss %.1e gradient %.1e (relative)" % (f, el, eg))
        assert el < 1e-12 and eg < 1e-12, "the sx stand-in disagrees with the sympy stand-in"
    for k, (name, T) in enumerate((("rocket", 50), ("quadrotor", 35), ("robotarm", 20))):
        r = run(name, 0.1, T, -1, None, seed=100 + k)
        np.savez_compressed(os.path.join(HERE, "ref_recmat_long_%s.npz" % name), **r)
        print("recmat", name, "T", T, "p", r["theta"].size, "loss", r["loss"], "|grad|", np.abs(r["grad"]).max())
