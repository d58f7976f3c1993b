"""GPU: ControlPlanning.step (Lagrange policy: the tile kernels; tanh-MLP [n, n]: the register kernel) and recmat_step (table policy on the size-generic adjoint kernel)
through the CLASS SURFACE run the reference's stored control / planning runs backwards (tests/undo_common.py) and reproduce the stored loss_trace of real CasADi runs:
Examples/OC/quadrotor/data/PDP_OC_results_trial_0.mat (uav_PDP.py), PDP_Recmat_results_trial_0.mat (uav_PDP_Recmat.py), OC/cartpole/data/PDP_Neural_trial_0.mat
(cartpole_PDP_neural.py), OC/robotarm/data/PDP_Neural_trial_0.mat and PDP_Recmat_results_trial_0.mat.  Stated tolerance: recovered loss 1e-11 relative (margins
recorded); the rollout of the stored final parameter reproduces the stored trajectory (1e-11) and cost (1e-12)."""
import os

import numpy as np
import pytest

from undo_common import CASES, final_parameter, undo

pytestmark = pytest.mark.gpu


def build_cp(name, g):
    from pdp_amd import JinEnv, PDP
    dt = float(g["dt"])
    env_kw = {k[4:]: float(g[k]) for k in g.files if k.startswith("env_")}
    if name.startswith("quadrotor"):
        env = JinEnv.Quadrotor()
        env.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01)                         # uav_PDP.py:9-14
        env.initCost(wr=1, wv=1, wq=5, ww=1, wthrust=0.1)
    elif name.startswith("cartpole"):
        env = JinEnv.CartPole()
        env.initDyn(mc=env_kw["mc"], mp=env_kw["mp"], l=env_kw["l"])
        env.initCost(wx=env_kw["wx"], wq=env_kw["wq"], wdx=env_kw["wdx"], wdq=env_kw["wdq"], wu=env_kw["wu"])
    else:
        env = JinEnv.RobotArm()
        env.initDyn(l1=env_kw["l1"], m1=env_kw["m1"], l2=env_kw["l2"], m2=env_kw["m2"], g=0)      # robotarm_PDP_Recmat.py:12
        env.initCost(wq1=env_kw["wq1"], wq2=env_kw["wq2"], wdq1=env_kw["wdq1"], wdq2=env_kw["wdq2"], wu=env_kw["wu"])
    cp = PDP.ControlPlanning("undo " + name)
    cp.setStateVariable(env.X)
    cp.setControlVariable(env.U)
    cp.setDyn(env.X + dt * env.f)
    cp.setPathCost(env.path_cost)
    cp.setFinalCost(env.final_cost)
    return cp


@pytest.mark.parametrize("name", CASES)
def test_class_surface_undoes_the_stored_gradient_steps(golden_dir, margins, name):
    g = np.load(os.path.join(golden_dir, "undo_%s.npz" % name))
    cp = build_cp(name, g)
    T, x0 = int(g["horizon"]), g["x0"]
    P = final_parameter(name, g)
    if name.endswith("recmat"):
        cp.recmat_init_step(T, -1)
        step = lambda th: cp.recmat_step(x0, T, th)
        sol = cp.recmat_unwarp(x0, T, P)
    else:
        if name.endswith("poly"):
            cp.init_step(T)
        else:
            cp.init_step_neural_policy([cp.n_state, cp.n_state])
        step = lambda th: cp.step(x0, T, th)
        sol = cp.integrateSys(x0, T, P)
    assert cp.n_auxvar == P.size
    cost = float(np.asarray(sol["cost"]).squeeze())
    margins.check("stored run %s: rollout of the stored final parameter, cost (relative)" % name, abs(cost - float(g["solved_cost"])) / abs(float(g["solved_cost"])), 1e-12)
    margins.check("stored run %s: rollout of the stored final parameter, state trajectory" % name, np.abs(sol["state_traj"] - g["solved_state"]).max(), 1e-11)
    for k, (got, stored, res) in enumerate(undo(step, P, float(g["lr"]), g["loss_tail"])):
        assert res <= 1e-12 * max(1.0, np.abs(P).max()), (name, k, res)
        margins.check("stored run %s: loss recovered %d gradient steps before the end vs stored loss_trace (relative)" % (name, k + 1), abs(got - stored) / abs(stored), 1e-11)
