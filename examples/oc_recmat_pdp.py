#!/usr/bin/env python3
"""Open-loop optimal control by PDP's recovery-matrix variant on the GPU - the reference's Examples/OC/rocket/rocket_PDP_Recmat.py:40-90 and
Examples/OC/quadrotor/uav_PDP_Recmat.py (also pendulum / robot arm): one parameter per control and time step (recmat_init_step(horizon, -1)), gradient descent with
recmat_step, the trajectory by recmat_unwarp, beside the "true" solution of OCSys.ocSolver; results saved with the reference's field names (loss_trace,
parameter_trace, learning_rate, solved_solution, true_solution, time_passed, dt, horizon).

Here recmat_step is ONE launch (PDP_POLICY_TABLE on the size-generic adjoint kernel) and the loop stays on the device: pdp_amd.irl.GDLoop replays step + update as a
hipGraph (--graph), or the host drives it call by call like the reference's script.

    python examples/oc_recmat_pdp.py --system rocket --iters 2000 --graph [--grid 10]     (--grid: time-grid cells instead of one per step)
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

# initial states and horizons of the reference's drivers
SETUP = {
    "rocket": dict(horizon=50, x0=[10, -8, 5.] + [-.1, 0.0, -0.0] + JinEnv.toQuaternion(1.5, [0, 0, 1]) + [0, -0.0, 0.0]),       # rocket_PDP_Recmat.py:16-29
    "quadrotor": dict(horizon=35, x0=[-8, -6, 9.] + [0.0, 0.0, 0.0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0.0, 0.0, 0.0]),    # uav_PDP_Recmat.py:25-32
    "pendulum": dict(horizon=20, x0=[0.0, 0.0]),
    "robotarm": dict(horizon=20, x0=[np.pi / 4, np.pi / 2, 0, 0]),
    "cartpole": dict(horizon=30, x0=[0.0, 0.0, 0.0, 0.0]),
}


def build(system):
    """(ControlPlanning object, OCSys object of the same problem, dt)"""
    env, dt = zoo.make_env(system, "oc")
    dyn = env.X + dt * env.f
    cp = PDP.ControlPlanning(system)                  # (same label and equations as the pre-built zoo model: no compilation at run time)
    cp.setStateVariable(env.X)
    cp.setControlVariable(env.U)
    cp.setDyn(dyn)
    cp.setPathCost(env.path_cost)
    cp.setFinalCost(env.final_cost)
    true_oc = PDP.OCSys(system + " true oc")
    true_oc.setStateVariable(env.X)
    true_oc.setControlVariable(env.U)
    true_oc.setDyn(dyn)
    true_oc.setPathCost(env.path_cost)
    true_oc.setFinalCost(env.final_cost)
    return cp, true_oc, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="rocket", choices=sorted(SETUP))
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--sigma", type=float, default=5.0, help="initial parameter = true optimal controls + sigma * N(0, 1) (rocket_PDP_Recmat.py:53)")
    ap.add_argument("--grid", type=int, default=-1, help="-1: one cell per time step (the reference's recmat_init_step(horizon, -1)); k > 0: k equal cells")
    ap.add_argument("--graph", action="store_true", help="keep the loop on the device: step + update replayed as one hipGraph per iteration (pdp_amd.irl.GDLoop)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()

    cp, true_oc, dt = build(a.system)
    horizon = a.horizon or SETUP[a.system]["horizon"]
    ini_state = np.array(SETUP[a.system]["x0"], dtype=float)
    true_sol = true_oc.ocSolver(ini_state=ini_state, horizon=horizon)
    print("true cost (OCSys.ocSolver): %.8f" % float(true_sol["cost"]))
    t0 = time.time()
    cp.recmat_init_step(horizon, -1 if a.grid < 0 else np.linspace(0, 1, a.grid + 1))
    rng = np.random.default_rng(a.seed)
    true_u = true_sol["control_traj_opt"]
    cell_start = cp.time_grid[:-1]
    theta = (true_u[cell_start] + a.sigma * rng.standard_normal((cp.whorizon, cp.n_control))).reshape(-1)
    parameter_trace, loss_trace = [theta.copy()], []
    if a.graph:
        from pdp_amd.irl import GDLoop
        loop = GDLoop(cp.warped_step_fn(ini_state), theta, a.lr, max_steps=a.iters)
        loop.run(a.iters)
        r = loop.results()
        loss_trace = list(r["loss_trace"])
        parameter_trace += list(r["parameter_trace"])
        theta = r["parameter_trace"][-1]
    else:
        for k in range(a.iters):
            loss, dp = cp.recmat_step(ini_state, horizon, theta)
            theta = theta - a.lr * dp
            loss_trace.append(loss)
            parameter_trace.append(theta.copy())
            if k % max(1, a.iters // 10) == 0:
                print("iter %6d  loss %.8f" % (k, loss))
    sol = cp.recmat_unwarp(ini_state, horizon, theta)
    dt_run = time.time() - t0
    save = {"trail_no": 0, "parameter_trace": parameter_trace, "loss_trace": loss_trace, "learning_rate": a.lr, "solved_solution": sol, "true_solution": true_sol,
            "time_passed": dt_run, "dt": dt, "horizon": horizon}
    if a.out:
        sio.savemat(a.out, {"results": save})
    print("done: %d iterations (p = %d) in %.2f s = %.0f iterations/s%s; loss %.6f -> %.6f (true optimum %.6f)" % (
        a.iters, cp.n_auxvar, dt_run, a.iters / dt_run, ", one hipGraph per iteration" if a.graph else "", loss_trace[0], loss_trace[-1], float(true_sol["cost"])))
    return loss_trace


if __name__ == "__main__":
    main()
