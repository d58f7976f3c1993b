#!/usr/bin/env python3
"""Inverse reinforcement learning with PDP on the GPU - the loop of the reference's Examples/IRL/<sys>/<sys>_PDP.py
(e.g. cartpole_PDP.py:32-94) with the per-demo Python loop replaced by ONE batched launch per stage:

    for k in iterations:   traj  = ocSolver(theta_k)                     (IPOPT's iteration in one GPU launch, started from the first-order PREDICTION
                                                                          traj_{k-1} + d traj / d theta (theta_k - theta_{k-1}) the previous gradient step provides)
                           loss, dp = aux system + Riccati + chain rule  (fused kernel; keeps d traj / d theta for the next prediction)
                           theta_{k+1} = theta_k - lr * mean(dp)

Demonstrations: the reference's stored demos (examples/data/demos_<sys>.npz, copies of the extracts under tests/golden), or any `<name>_demos.mat` written by the reference's
generate_demos.py (field names trajectories[i].state_traj_opt / control_traj_opt, true_parameter, dt; e.g.
Examples/IRL/cartpole/generate_demos.py:38-43) through --demos.  Results are saved with the reference's field names
(results.loss_trace / parameter_trace / learning_rate / time_passed) so its plotting scripts keep working.

    python examples/irl_pdp.py --system cartpole --iters 200 --lr 1e-4 [--demos path/to/cartpole_demos.mat]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pdp_amd import PDP, ocsolver, zoo          # noqa: E402
from pdp_amd.sx import vertcat                  # noqa: E402


def load_demos(path):
    """(state [B,T+1,n], control [B,T,m], true_parameter [p]) from the npz fixtures or from a .mat in the reference's schema"""
    if path.endswith(".npz"):
        d = np.load(path)
        return d["state"], d["control"], d["true_parameter"]
    d = sio.loadmat(path)
    tr = d["trajectories"]
    xs = np.stack([np.asarray(tr[0, i]["state_traj_opt"][0, 0], dtype=float) for i in range(tr.shape[1])])
    us = np.stack([np.asarray(tr[0, i]["control_traj_opt"][0, 0], dtype=float) for i in range(tr.shape[1])])
    return xs, us, d["true_parameter"].astype(float).flatten()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="cartpole", choices=["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--sigma", type=float, default=0.3, help="initial parameter = true + U(-sigma/2, sigma/2)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--init", default=None, help="initial parameter: comma-separated values or a .npy file (default: true + U(-sigma/2, sigma/2)); e.g. row 0 of a stored "
                                                 "parameter_trace to replay the reference's run")
    ap.add_argument("--out", default=None)
    ap.add_argument("--record", default="full", choices=["full", "primal"],
                    help="what the gradient unit keeps for the next solve's predicted start: states, controls and multipliers (full) or states and controls only "
                         "(primal: cheaper to write and read; enough where the multipliers move little per step, e.g. the quadrotor)")
    ap.add_argument("--graph", action="store_true",
                    help="keep the loop on the device and replay each iteration as one hipGraph (pdp_amd.irl.IRLLoop): no host work per iteration, traces written by the graph")
    ap.add_argument("--demos", default=None, help="<name>_demos.mat in the reference's schema (default: the stored demos of --system)")
    a = ap.parse_args()

    env, dt = zoo.make_env(a.system, "irl")
    oc = PDP.OCSys(a.system)
    oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar))
    oc.setControlVariable(env.U)
    oc.setStateVariable(env.X)
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    oc.diffPMP()

    demo_x, demo_u, true_parameter = load_demos(a.demos or os.path.join(ROOT, "examples", "data", "demos_%s.npz" % a.system))
    T = demo_u.shape[1]
    rng = np.random.default_rng(a.seed)
    theta = true_parameter + a.sigma * rng.random(true_parameter.size) - a.sigma / 2
    if a.init is not None:
        theta = np.load(a.init).astype(float).reshape(-1) if a.init.endswith(".npy") else np.array([float(v) for v in a.init.split(",")])
        assert theta.size == true_parameter.size, "--init: %d values for %d parameters" % (theta.size, true_parameter.size)
    loss_trace, parameter_trace = [], []
    warm, predict, theta_prev = None, None, None
    fused = oc.model().n <= 16 and oc.model().m <= 4 and oc.model().m + oc.model().p <= 16      # the kernels that keep the sensitivities
    t0 = time.time()
    if a.graph:
        assert fused, "--graph needs the fused kernels (n <= 16, m <= 4, m + p <= 16)"
        from pdp_amd.irl import IRLLoop
        loop = IRLLoop(oc.model(), demo_x, demo_u, theta, a.lr, record=a.record, max_steps=a.iters + 8)
        loop.run(a.iters)
        r = loop.results()
        loss_trace, parameter_trace = list(r["loss_trace"][:a.iters]), list(r["parameter_trace"][:a.iters])
        if r["unconverged_solves"] or r["riccati_trouble"]:
            print("warning: %d OC solves did not converge, %d trajectories with numerical trouble in the Riccati sweep" % (r["unconverged_solves"], r["riccati_trouble"]))
        for k in range(0, a.iters, max(1, a.iters // 10)):
            print("iter %5d  loss %.6e  |theta - theta*| %.4f" % (k, loss_trace[k], np.abs(parameter_trace[k] - true_parameter).max()))
        save = {"trail_no": 0, "loss_trace": loss_trace, "parameter_trace": parameter_trace, "learning_rate": a.lr, "time_passed": time.time() - t0}
        if a.out:
            sio.savemat(a.out, {"results": save})
        print("done: %d iterations x %d demos in %.2f s, one hipGraph per iteration  (loss %.4e -> %.4e)" % (a.iters, demo_x.shape[0], save["time_passed"], loss_trace[0], loss_trace[-1]))
        return loss_trace
    for k in range(a.iters):
        # first iterate: cold, the reference's all-zero guess (PDP.py:155,166).  Afterwards the multiple-shooting solver starts from the previous solution moved
        # along its own sensitivities, (x, u, lambda)_{k-1} + (X, U, Lambda)_{k-1} (theta_k - theta_{k-1}) - the auxiliary control system the gradient step has just
        # solved IS that derivative (PDP.py:582-608) - and needs about one Newton iteration (two from the unmoved previous solution)
        if predict is not None:
            predict["dtheta"] = theta - theta_prev
        sol = ocsolver.solve_batch(oc, demo_x[:, 0], T, theta, warm_start=warm, predict=predict)
        if not bool(sol["converged"].all()):
            print("iter %5d  warning: %d of %d OC solves did not converge" % (k, int((~sol["converged"]).sum()), demo_x.shape[0]))
        warm = {key: sol[key] for key in ("state", "control", "costate")}
        out = oc.pdp_grad_batch(sol["control"], theta, demo_x, demo_u, state_traj=sol["state"], costate_traj=sol["costate"], want_predict_record=(("primal" if a.record == "primal" else True) if fused else False))
        if fused:
            predict, theta_prev = {"record": out["predict_record"], "primal": a.record == "primal"}, theta.copy()
        if int(out["status"].sum()) != 0:
            print("iter %5d  warning: Riccati sweep reported numerical trouble on %d trajectories" % (k, int((out["status"] != 0).sum())))
        loss = float(out["loss"].mean())
        dp = out["grad"].mean(dim=0).cpu().numpy()
        theta = theta - a.lr * dp
        loss_trace.append(loss)
        parameter_trace.append(theta.copy())
        if k % max(1, a.iters // 10) == 0:
            print("iter %5d  loss %.6e  |theta - theta*| %.4f" % (k, loss, np.abs(theta - true_parameter).max()))
    save = {"trail_no": 0, "loss_trace": loss_trace, "parameter_trace": parameter_trace, "learning_rate": a.lr, "time_passed": time.time() - t0}
    if a.out:
        sio.savemat(a.out, {"results": save})
    print("done: %d iterations x %d demos in %.2f s  (loss %.4e -> %.4e)" % (a.iters, demo_x.shape[0], save["time_passed"], loss_trace[0], loss_trace[-1]))
    return loss_trace


if __name__ == "__main__":
    main()
