"""CPU: the oracle runs the reference's stored control / planning runs backwards (tests/undo_common.py): ControlPlanningOracle.step with the Lagrange and the tanh-MLP
policy and the recovery-matrix gradient reproduce the stored loss_trace of real CasADi runs from the stored final parameter - the oracle's GRADIENTS for these paths
are pinned on reference-held data (SURVEY.md section 8c listed them as "pinned analytically" only)."""
import os

import numpy as np
import pytest

from undo_common import BACK, CASES, final_parameter, undo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_cp(name, g):
    from oracle import models, pdp_oracle as po
    dt = float(g["dt"])
    if name.startswith("quadrotor"):
        return po.make_cp(models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1), dt)       # uav_PDP.py:9-14
    env = {k[4:]: float(g[k]) for k in g.files if k.startswith("env_")}
    if name.startswith("cartpole"):
        return po.make_cp(models.cart_pole(**env), dt)
    return po.make_cp(models.robot_arm(g=0, **env), dt)                                                                             # robotarm_PDP_Recmat.py:12: g = 0


def recmat_step(cp, x0, T):
    """loss and d loss / d u_t of the open-loop problem by the PMP costates (what recmat_step's recovery matrix evaluates, PDP.py:1039-1114), in numpy"""
    n, m = cp.n, cp.m

    def step(theta):
        u = np.asarray(theta, float).reshape(T, m)
        xs = np.zeros((T + 1, n))
        xs[0] = x0
        c = 0.0
        for t in range(T):
            xs[t + 1] = np.asarray(cp.dyn_fn(xs[t], u[t]), float).reshape(-1)
            c += float(cp.path_cost_fn(xs[t], u[t]))
        c += float(cp.final_cost_fn(xs[T]))
        lam = np.asarray(cp.dhx_fn(xs[T]), float).reshape(-1)
        gr = np.zeros_like(u)
        for t in range(T - 1, -1, -1):
            F, G = np.asarray(cp.dfx_fn(xs[t], u[t]), float).reshape(n, n), np.asarray(cp.dfu_fn(xs[t], u[t]), float).reshape(n, m)
            gr[t] = np.asarray(cp.dcu_fn(xs[t], u[t]), float).reshape(-1) + G.T @ lam
            lam = np.asarray(cp.dcx_fn(xs[t], u[t]), float).reshape(-1) + F.T @ lam
        return c, gr.reshape(-1)
    return step


@pytest.mark.parametrize("name", CASES)
def test_oracle_undoes_the_stored_gradient_steps(name):
    g = np.load(os.path.join(GOLDEN, "undo_%s.npz" % name))
    cp = oracle_cp(name, g)
    T, x0 = int(g["horizon"]), g["x0"]
    P = final_parameter(name, g)
    if name.endswith("recmat"):
        step = recmat_step(cp, x0, T)
    else:
        if name.endswith("poly"):
            cp.init_step(T)
        else:
            cp.init_step_neural_policy([cp.n, cp.n])
        assert cp.n_auxvar == P.size
        step = lambda th: cp.step(x0, T, th)
        sol = cp.integrateSys(x0, T, P)
        assert abs(sol["cost"] - float(g["solved_cost"])) <= 1e-12 * abs(float(g["solved_cost"])) and np.abs(sol["state_traj"] - g["solved_state"]).max() <= 1e-11
    for k, (got, stored, res) in enumerate(undo(step, P, float(g["lr"]), g["loss_tail"])):
        assert res <= 1e-12 * max(1.0, np.abs(P).max()), (name, k, res)
        assert abs(got - stored) <= 1e-11 * abs(stored), (name, k, got, stored)
