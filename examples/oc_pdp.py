#!/usr/bin/env python3
"""Control / planning with PDP on the GPU - the loop of the reference's Examples/OC/<sys>/*_PDP.py (e.g. quadrotor/uav_PDP.py:52-83):
gradient descent on the policy parameters with ControlPlanning.step, here for a BATCH of initial states in one launch per
iteration (the reference optimises one initial state at a time).  --policy poly (Lagrange, init_step) or mlp (init_step_neural_policy).

    python examples/oc_pdp.py --system quadrotor --horizon 35 --iters 500 --batch 64
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pdp_amd import PDP, JinEnv, zoo          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="quadrotor", choices=["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
    ap.add_argument("--horizon", type=int, default=35)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--policy", default="poly", choices=["poly", "mlp"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    env, dt = zoo.make_env(a.system, "oc")
    oc = PDP.ControlPlanning(a.system)
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U)
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    if a.policy == "poly":
        oc.init_step(a.horizon)
    else:
        oc.init_step_neural_policy()
    rng = np.random.default_rng(a.seed)
    n = oc.n_state
    x0 = 0.1 * rng.standard_normal((a.batch, n))
    if a.system in ("quadrotor", "rocket"):
        x0[:, :3] = rng.uniform(-4, 4, (a.batch, 3)) + np.array([0, 0, 6.0])
        x0[:, 3:6] = 0
        x0[:, 6:10] = JinEnv.toQuaternion(0, [1, -1, 1])
        x0[:, 10:] = 0
    theta = (1.0 if a.policy == "poly" else 0.1) * rng.standard_normal((a.batch, oc.n_auxvar))     # one policy per initial state
    loss_trace = []
    t0 = time.time()
    for k in range(a.iters):
        loss, dp = oc.step_batch(x0, a.horizon, theta)
        theta = theta - a.lr * dp.cpu().numpy()
        loss_trace.append(float(loss.mean()))
        if k % max(1, a.iters // 10) == 0:
            print("iter %5d  mean loss %.6e" % (k, loss_trace[-1]))
    sol = oc.integrateSys(x0[0], a.horizon, theta[0])
    # the "true" solution the reference stores beside it (Examples/OC/quadrotor/uav_PDP.py:34-41, 76-83): OCSys.ocSolver on the same problem from the same initial state
    true_oc = PDP.OCSys(a.system + " true oc")
    true_oc.setStateVariable(env.X)
    true_oc.setControlVariable(env.U)
    true_oc.setDyn(env.X + dt * env.f)
    true_oc.setPathCost(env.path_cost)
    true_oc.setFinalCost(env.final_cost)
    true_sol = true_oc.ocSolver(ini_state=x0[0], horizon=a.horizon)
    save = {"trail_no": 0, "parameter_trace": [theta[0]], "loss_trace": loss_trace, "learning_rate": a.lr, "solved_solution": sol, "true_solution": true_sol,
            "time_passed": time.time() - t0, "dt": dt, "horizon": a.horizon}
    if a.out:
        sio.savemat(a.out, {"results": save})
    print("done: %d iterations x %d initial states in %.2f s (%.0f trajectory-iterations/s); mean loss %.4e -> %.4e" % (
        a.iters, a.batch, save["time_passed"], a.iters * a.batch / save["time_passed"], loss_trace[0], loss_trace[-1]))
    return loss_trace


if __name__ == "__main__":
    main()
