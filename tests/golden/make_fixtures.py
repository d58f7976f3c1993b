#!/usr/bin/env python3
"""Extract the reference's stored results (.mat, written by real CasADi+IPOPT runs on the author's
machine) into small .npz fixtures.  Runs ONLY in the build container (needs /root/reference); the
.npz files are committed, the reference never travels.  Data only - no reference source is copied.

Sources (all under /root/reference/Examples):
  IRL/<sys>/data/<name>_demos.mat            -> demos_<sys>.npz   (x*,u*,lambda*,cost,theta*,dt)
  IRL/<sys>/data/PDP_results_trial_0.mat     -> irltrace_<sys>.npz (selected rows of the GD trace), irltrace_head_<sys>.npz (its first 202 rows, consecutive)
  SysID/<sys>/data/<name>_iodata.mat         -> iodata_<sys>.npz
  OC/quadrotor/data/PDP_OC_results_trial_0   -> oc_quadrotor.npz  (solved_solution / true_solution)
  OC/cartpole/data/PDP_Neural_trial_0        -> oc_cartpole_neural.npz (final MLP params + rollout)
  OC/{quadrotor,cartpole,robotarm}/data/PDP_{OC,Recmat,Neural}_*trial_0 -> undo_<name>.npz (final parameter / controls, last losses, settings: undo_fixtures)
"""
import os
import sys
import numpy as np
import scipy.io as sio

REF = os.environ.get("PDP_REFERENCE", "/root/reference")
EX = os.path.join(REF, "Examples")
OUT = os.path.dirname(os.path.abspath(__file__))

HEAD_ROWS = 202   # rows 0..201 of parameter_trace / loss_trace: 200 consecutive gradient-descent iterations

SYS = {  # system -> (demo file stem, iodata stem)
    "pendulum": ("pendulum", "pendulum"),
    "cartpole": ("cartpole", "cartpole"),
    "robotarm": ("robotarm", "robotarm"),
    "quadrotor": ("uav", "uav"),
    "rocket": ("rocket", "rocket"),
}


def demos(sysname, stem):
    d = sio.loadmat(os.path.join(EX, "IRL", sysname, "data", stem + "_demos.mat"))
    tr = d["trajectories"]
    n_demo = tr.shape[1]
    xs = np.stack([tr[0, i]["state_traj_opt"][0, 0] for i in range(n_demo)])
    us = np.stack([tr[0, i]["control_traj_opt"][0, 0] for i in range(n_demo)])
    ls = np.stack([tr[0, i]["costate_traj_opt"][0, 0] for i in range(n_demo)])
    cost = np.array([float(tr[0, i]["cost"][0, 0].squeeze()) for i in range(n_demo)])
    np.savez_compressed(os.path.join(OUT, "demos_%s.npz" % sysname), state=xs, control=us, costate=ls, cost=cost,
                        true_parameter=d["true_parameter"].astype(float).flatten(), dt=float(d["dt"].squeeze()))
    print("demos", sysname, xs.shape, us.shape, ls.shape, cost)


def irltrace(sysname):
    f = os.path.join(EX, "IRL", sysname, "data", "PDP_results_trial_0.mat")
    r = sio.loadmat(f)["results"][0, 0]
    P = np.asarray(r["parameter_trace"], dtype=float)
    K = P.shape[0]
    P = P.reshape(K, -1)
    L = np.asarray(r["loss_trace"], dtype=float).flatten()
    lr = float(r["learning_rate"].squeeze())
    # identities: loss(P[k]) == L[k+1] ; grad(P[k]) == (P[k]-P[k+1])/lr
    ks = sorted(set(k for k in [0, 1, 2, 100, 500, 1000, 2000, 5000, K - 3, K - 2] if 0 <= k <= K - 2))
    np.savez_compressed(os.path.join(OUT, "irltrace_%s.npz" % sysname), k=np.array(ks), param=P[ks], param_next=P[[k + 1 for k in ks]],
                        loss_next=L[[k + 1 for k in ks]], lr=lr, K=K, time_passed=float(r["time_passed"].squeeze()))
    print("irltrace", sysname, "K", K, "p", P.shape[1], "lr", lr, "rows", ks)
    # the head of the stored gradient-descent trace, every row: what a replay of the reference's loop (Examples/IRL/<sys>/<sys>_PDP.py) from P[0] must reproduce
    H = min(HEAD_ROWS, K)
    np.savez_compressed(os.path.join(OUT, "irltrace_head_%s.npz" % sysname), param=P[:H], loss=L[:H], lr=lr, K=K)


def iodata(sysname, stem):
    x = sio.loadmat(os.path.join(EX, "SysID", sysname, "data", stem + "_iodata.mat"))[stem + "_iodata"][0, 0]
    np.savez_compressed(os.path.join(OUT, "iodata_%s.npz" % sysname), inputs=np.asarray(x["batch_inputs"], float),
                        states=np.asarray(x["batch_states"], float), true_parameter=np.asarray(x["true_parameter"], float).flatten())
    print("iodata", sysname, x["batch_inputs"].shape, x["batch_states"].shape, x["true_parameter"].flatten())


def oc_quadrotor():
    r = sio.loadmat(os.path.join(EX, "OC", "quadrotor", "data", "PDP_OC_results_trial_0.mat"))["results"][0, 0]
    ss = r["solved_solution"][0, 0]
    ts = r["true_solution"][0, 0]
    np.savez_compressed(os.path.join(OUT, "oc_quadrotor.npz"), state=ss["state_traj"], control=ss["control_traj"], cost=float(ss["cost"].squeeze()),
                        true_state=ts["state_traj_opt"], true_control=ts["control_traj_opt"], true_costate=ts["costate_traj_opt"],
                        true_cost=float(ts["cost"].squeeze()), dt=float(r["dt"].squeeze()), horizon=int(r["horizon"].squeeze()),
                        time_passed=float(r["time_passed"].squeeze()), n_iter=int(np.asarray(r["loss_trace"]).size))
    print("oc_quadrotor", ss["state_traj"].shape, float(ss["cost"].squeeze()), float(ts["cost"].squeeze()))


def oc_cartpole_neural():
    r = sio.loadmat(os.path.join(EX, "OC", "cartpole", "data", "PDP_Neural_trial_0.mat"))["results"][0, 0]
    P = np.asarray(r["parameter_trace"], float)
    P = P.reshape(P.shape[0], -1)
    ss = r["solved_solution"][0, 0]
    keys = r.dtype.names
    extra = {}
    for k in ("dt", "horizon"):
        if k in keys:
            extra[k] = float(np.asarray(r[k]).squeeze())
    np.savez_compressed(os.path.join(OUT, "oc_cartpole_neural.npz"), param_final=P[-1], state=ss["state_traj"], control=ss["control_traj"],
                        cost=float(ss["cost"].squeeze()), loss_last=float(np.asarray(r["loss_trace"]).flatten()[-1]), **extra)
    print("oc_cartpole_neural", P.shape, ss["state_traj"].shape, keys)


def undo_fixtures():
    """The tails of the reference's stored CONTROL / PLANNING runs (Examples/OC/*/data): final parameter (or the final controls, from which the Lagrange parameter follows by
    a least-squares fit), learning rate, the last losses, and the settings the run was made with.  The drivers update `current_parameter -= lr * dp` in place and append the
    same array to parameter_trace every iteration, so only the FINAL parameter survives in the files - but the loop can be run BACKWARDS from it: P_{k} solves
    P_k - lr * grad(P_k) = P_{k+1}, and loss(P_k) must be the stored loss_trace[k].  That pins ControlPlanning.step / recmat_step GRADIENTS on outputs of real CasADi runs
    (tests/test_undo_stored_runs.py, tests/test_gpu_undo_stored_runs.py); SURVEY.md section 8c knew of no reference-held vector for them."""
    NL = 12

    def st(s):
        s = s[0, 0]
        return {k: np.asarray(s[k], dtype=float).squeeze() if s[k].dtype != object else np.asarray(s[k][0, 0], dtype=float) for k in s.dtype.names}

    def common(r):
        L = np.asarray(r["loss_trace"], float).flatten()
        ss = st(r["solved_solution"])
        return dict(lr=float(r["learning_rate"].squeeze()), K=L.size, loss_tail=L[-NL:], dt=float(r["dt"].squeeze()), horizon=int(r["horizon"].squeeze()),
                    x0=ss["state_traj"][0], solved_state=ss["state_traj"], solved_control=ss["control_traj"].reshape(ss["state_traj"].shape[0] - 1, -1),
                    solved_cost=float(ss["cost"]))

    def final_param(r):
        P = np.asarray(r["parameter_trace"], float)
        return P.reshape(P.shape[0], -1)[-1]
    jobs = [("quadrotor_recmat", "OC/quadrotor/data/PDP_Recmat_results_trial_0.mat", None, False), ("quadrotor_poly", "OC/quadrotor/data/PDP_OC_results_trial_0.mat", None, False),
            ("cartpole_neural", "OC/cartpole/data/PDP_Neural_trial_0.mat", "cartpole", True), ("robotarm_neural", "OC/robotarm/data/PDP_Neural_trial_0.mat", "robotarm", True),
            ("robotarm_recmat", "OC/robotarm/data/PDP_Recmat_results_trial_0.mat", "robotarm", False)]
    for name, rel, envkey, has_param in jobs:
        r = sio.loadmat(os.path.join(EX, rel))["results"][0, 0]
        d = common(r)
        if envkey:
            e = r[envkey][0, 0]
            d.update({"env_" + k: float(np.asarray(e[k]).squeeze()) for k in e.dtype.names})
        if has_param:
            d["final_parameter"] = final_param(r)
        np.savez_compressed(os.path.join(OUT, "undo_%s.npz" % name), **d)
        print("undo", name, "K", d["K"], "lr", d["lr"], "T", d["horizon"], "tail", d["loss_tail"][-2:], "cost", d["solved_cost"])


if __name__ == "__main__":
    if not os.path.isdir(EX):
        sys.exit("reference not found at %s (fixtures are generated in the build container only)" % REF)
    for s, (dstem, istem) in SYS.items():
        demos(s, dstem)
        irltrace(s)
        iodata(s, istem)
    oc_quadrotor()
    oc_cartpole_neural()
    undo_fixtures()
