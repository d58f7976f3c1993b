#!/usr/bin/env python3
"""System identification with PDP on the GPU - the loop of the reference's Examples/SysID/<sys>/*_PDP.py (quadrotor/uav_PDP.py:33-59)
on the reference's stored input/state data (tests/golden/iodata_<sys>.npz), or any `<stem>_iodata.mat` written by the reference's
generate_traj.py (struct <stem>_iodata with batch_inputs / batch_states / true_parameter, e.g.
Examples/SysID/quadrotor/generate_traj.py:36-42) through --data.

    python examples/sysid_pdp.py --system quadrotor --iters 2000 --lr 1e-4 [--data path/to/uav_iodata.mat]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.io as sio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pdp_amd import PDP, zoo          # noqa: E402


def load_iodata(path):
    """(inputs [B,T,m], states [B,T+1,n], true_parameter [p]) from the npz fixtures or from a .mat in the reference's schema"""
    if path.endswith(".npz"):
        io = np.load(path)
        return io["inputs"], io["states"], io["true_parameter"]
    d = sio.loadmat(path)
    key = [k for k in d if k.endswith("_iodata")][0]
    x = d[key][0, 0]
    return np.asarray(x["batch_inputs"], float), np.asarray(x["batch_states"], float), np.asarray(x["true_parameter"], float).flatten()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--system", default="quadrotor", choices=["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--sigma", type=float, default=0.6)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--graph", action="store_true", help="keep the loop on the device (pdp_amd.irl.GDLoop: two launches per iteration, replayed as a hipGraph); equal horizons only")
    ap.add_argument("--data", default=None, help="<stem>_iodata.mat in the reference's schema (default: the stored data of --system)")
    a = ap.parse_args()
    env, dt = zoo.make_env(a.system, "sysid")
    sid = PDP.SysID(a.system)
    sid.setAuxvarVariable(env.dyn_auxvar)
    sid.setStateVariable(env.X)
    sid.setControlVariable(env.U)
    sid.setDyn(env.X + dt * env.f)
    inputs, states, true_parameter = load_iodata(a.data or os.path.join(ROOT, "tests", "golden", "iodata_%s.npz" % a.system))
    batch_inputs = [inputs[i] for i in range(inputs.shape[0])]
    batch_states = [states[i] for i in range(states.shape[0])]
    rng = np.random.default_rng(a.seed)
    theta = true_parameter + a.sigma * rng.random(true_parameter.size) - a.sigma / 2
    loss_trace, parameter_trace = [], []
    t0 = time.time()
    if a.graph:
        from pdp_amd import runtime as rt
        from pdp_amd.irl import GDLoop
        mdl = sid.model()
        u_d, x_d = rt.dev(np.stack(batch_inputs)), rt.dev(np.stack(batch_states))
        loop = GDLoop(lambda th: mdl.sysid_step(u_d, x_d, th), theta, a.lr, max_steps=a.iters + 8)
        loop.run(a.iters)
        r = loop.results()
        loss_trace, parameter_trace = list(r["loss_trace"][:a.iters]), list(r["parameter_trace"][:a.iters])
        theta = parameter_trace[-1]
        for k in range(0, a.iters, max(1, a.iters // 10)):
            print("iter %5d  loss %.6e  theta %s" % (k, loss_trace[k], np.array2string(parameter_trace[k], precision=4)))
    for k in range(0 if not a.graph else a.iters, a.iters):
        loss, dp = sid.step(batch_inputs, batch_states, theta)
        theta = theta - a.lr * dp
        loss_trace.append(loss)
        parameter_trace.append(theta.copy())
        if k % max(1, a.iters // 10) == 0:
            print("iter %5d  loss %.6e  theta %s" % (k, loss, np.array2string(theta, precision=4)))
    save = {"trail_no": 0, "loss_trace": loss_trace, "parameter_trace": parameter_trace, "learning_rate": a.lr, "time_passed": time.time() - t0}
    if a.out:
        sio.savemat(a.out, {"results": save})
    print("done: %d iterations in %.2f s; loss %.4e -> %.4e; |theta - theta*| = %.4f" % (a.iters, save["time_passed"], loss_trace[0], loss_trace[-1],
                                                                                         np.abs(theta - true_parameter).max()))
    return loss_trace


if __name__ == "__main__":
    main()
