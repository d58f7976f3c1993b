"""GPU: the example drivers (counterparts of the reference's Examples/{IRL,OC,SysID}/.../*_PDP.py loops) run end to end and learn."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + list(args), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


def test_irl_example_decreases_loss(tmp_path):
    out = run("irl_pdp.py", "--system", "cartpole", "--iters", "30", "--lr", "1e-4", "--out", str(tmp_path / "r.mat"))
    import scipy.io as sio
    r = sio.loadmat(str(tmp_path / "r.mat"))["results"][0, 0]
    L = r["loss_trace"].flatten()
    assert L.size == 30 and L[-1] < 0.5 * L[0], out
    assert r["parameter_trace"].shape[0] == 30 and float(r["learning_rate"].squeeze()) == 1e-4


def test_oc_example_decreases_loss():
    out = run("oc_pdp.py", "--system", "quadrotor", "--horizon", "35", "--iters", "60", "--batch", "8")
    first = float(out.split("mean loss")[1].split()[0])
    last = float(out.strip().split("->")[-1])
    assert last < first, out


def test_sysid_example_recovers_parameter():
    out = run("sysid_pdp.py", "--system", "robotarm", "--iters", "400", "--lr", "1e-4", "--sigma", "0.3")
    assert "done:" in out
    first, last = [float(v) for v in out.split("loss ")[-1].split(";")[0].split(" -> ")]
    assert last < 0.1 * first, out


def test_multi_gpu_example_runs_single_process():
    out = run("irl_pdp_multi_gpu.py", "--batch", "512", "--iters", "6")
    assert "done: 6 iterations x 512 trajectories on 1 GPU(s)" in out
    first = float(out.split("mean loss")[1].split()[0])
    assert np.isfinite(first)


def test_graphed_irl_loop_is_the_eager_loop():
    """pdp_amd.irl.IRLLoop: the iteration replayed as a hipGraph writes the traces the eager iterations write (same kernels, same buffers: bit for bit), both follow the
    host-driven loop of examples/irl_pdp.py (ocsolver.solve_batch + pdp_grad_batch, mean gradient on the host) to rounding, and nothing in the loop failed"""
    import torch
    sys.path.insert(0, ROOT)
    from pdp_amd import ocsolver, zoo
    from pdp_amd.irl import IRLLoop
    d = np.load(os.path.join(ROOT, "examples", "data", "demos_cartpole.npz"))
    dx, du, th_true = d["state"], d["control"], d["true_parameter"]
    mdl = zoo.get("cartpole", "irl")
    theta0 = th_true + 0.1 * np.array([1, -1, 1, -1, 1, -1, 1.0])[:th_true.size]
    n_it, lr = 14, 1e-4
    runs = {}
    for kind in ("graph", "eager"):
        for record in ("full", "primal"):
            loop = IRLLoop(mdl, dx, du, theta0, lr, record=record, max_steps=64)
            loop.run(n_it, graphed=(kind == "graph"))
            r = loop.results()
            assert r["iterations"] >= n_it and r["unconverged_solves"] == 0 and r["riccati_trouble"] == 0
            runs[kind, record] = r
    for record in ("full", "primal"):
        g, e = runs["graph", record], runs["eager", record]
        assert np.array_equal(g["loss_trace"][:n_it], e["loss_trace"][:n_it]) and np.array_equal(g["parameter_trace"][:n_it], e["parameter_trace"][:n_it])
    # the host-driven loop (what examples/irl_pdp.py does without --graph), plain warm starts: same optimum of every solve to the solver's tolerance
    T = du.shape[1]
    theta, warm, trace = theta0.copy(), None, []
    for k in range(n_it):
        sol = mdl.oc_solve_ms(dx[:, 0], theta, T, tol=1e-10, warm=warm)
        assert bool(sol["converged"].all())
        warm = (sol["state"], sol["control"], sol["costate"])
        out = mdl.oc_pdp_grad(sol["control"], theta, dx, du, x=sol["state"], lam=sol["costate"])
        theta = theta - lr * out["grad"].mean(dim=0).cpu().numpy()
        trace.append((float(out["loss"].mean()), theta.copy()))
    for record in ("full", "primal"):
        g = runs["graph", record]
        assert np.allclose(g["loss_trace"][:n_it], [a for a, _ in trace], rtol=1e-8, atol=0)
        assert np.abs(g["parameter_trace"][:n_it] - np.array([b for _, b in trace])).max() <= 1e-9
    assert g["loss_trace"][n_it - 1] < g["loss_trace"][0]


def test_irl_example_with_graph_option(tmp_path):
    out = run("irl_pdp.py", "--system", "cartpole", "--iters", "30", "--lr", "1e-4", "--graph", "--out", str(tmp_path / "g.mat"))
    import scipy.io as sio
    r = sio.loadmat(str(tmp_path / "g.mat"))["results"][0, 0]
    L = r["loss_trace"].flatten()
    assert L.size == 30 and L[-1] < 0.5 * L[0], out
    assert r["parameter_trace"].shape == (30, 7)


@pytest.mark.parametrize("B,p", [(1, 1), (5, 7), (1024, 9), (8192, 420), (300, 1023), (3, 1024), (70, 5316)])
def test_one_launch_parameter_update(B, p):
    """pdp_gd_update_batched against numpy: batch means, theta <- theta - lr * mean gradient, traces at the device-side counter, health counters; strided gradient
    rows (the packed [B, p + 1] output); rows beyond the trace length are dropped; parameter vectors beyond one workgroup's width (p + 1 > 1024: a [64, 64] policy has
    5316 parameters) take the wide kernel"""
    import torch
    sys.path.insert(0, ROOT)
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(B + p)
    packed = torch.as_tensor(rng.standard_normal((B, p + 1)), device="cuda")
    loss = packed[:, p].contiguous()
    status = torch.as_tensor((rng.random(B) < 0.1).astype(np.int32) * 3, device="cuda")
    conv = torch.as_tensor((rng.random(B) < 0.8).astype(np.int32), device="cuda")
    iters = torch.as_tensor(rng.integers(0, 9, B).astype(np.int32), device="cuda")
    theta = torch.as_tensor(rng.standard_normal(p), device="cuda")
    th0 = theta.cpu().numpy().copy()
    dth = torch.zeros(p, dtype=torch.float64, device="cuda")
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    ltr, ptr_ = torch.zeros(2, dtype=torch.float64, device="cuda"), torch.zeros(2, p, dtype=torch.float64, device="cuda")
    lr = 0.37
    G = packed.cpu().numpy()[:, :p]
    want_d = -lr * G.mean(axis=0)
    for k in range(3):                      # the third call finds the traces full
        rt.gd_update(loss, packed[:, :p], lr, theta, dth, cnt, status=status, converged=conv, iterations=iters, loss_trace=ltr, parameter_trace=ptr_)
        sc = max(1e-300, np.abs(want_d).max())
        assert np.abs(dth.cpu().numpy() - want_d).max() <= 1e-13 * max(sc, lr * np.abs(G).max())
        assert np.abs(theta.cpu().numpy() - (th0 + (k + 1) * want_d)).max() <= 1e-12 * max(1.0, np.abs(th0).max())
    c = cnt.cpu().numpy()
    assert c[0] == 3 and c[1] == 3 * int((conv == 0).sum()) and c[2] == 3 * int((status != 0).sum()) and c[3] == 3 * int(iters.sum())
    assert np.abs(ltr.cpu().numpy() - float(loss.mean())).max() <= 1e-13 * max(1.0, abs(float(loss.mean())))
    assert np.abs(ptr_.cpu().numpy()[1] - (th0 + 2 * want_d)).max() <= 1e-12 * max(1.0, np.abs(th0).max())
    rt.gd_update(loss, packed[:, :p], lr, theta, dth, cnt)          # everything optional left out
    assert int(cnt[0]) == 4


def test_device_resident_gd_loop_for_sysid():
    """pdp_amd.irl.GDLoop around SysID.step (the loop of Examples/SysID/*/..._PDP.py: loss, dp = step(...); parameter -= lr * dp): graph replays = eager iterations bit for
    bit, both = the host-driven loop through PDP.SysID.step to rounding, and the parameter moves towards the true one"""
    sys.path.insert(0, ROOT)
    from pdp_amd import PDP, runtime as rt, zoo
    from pdp_amd.irl import GDLoop
    d = np.load(os.path.join(ROOT, "tests", "golden", "iodata_robotarm.npz"))
    inputs, states, th_true = d["inputs"], d["states"], d["true_parameter"]
    env, dt = zoo.make_env("robotarm", "sysid")
    sid = PDP.SysID("robotarm")
    sid.setAuxvarVariable(env.dyn_auxvar); sid.setStateVariable(env.X); sid.setControlVariable(env.U); sid.setDyn(env.X + dt * env.f)
    mdl = sid.model()
    theta0 = th_true + 0.1 * np.cos(np.arange(th_true.size))
    u_d, x_d = rt.dev(inputs), rt.dev(states)
    n_it, lr = 25, 1e-4
    runs = {}
    for kind in ("graph", "eager"):
        loop = GDLoop(lambda th: mdl.sysid_step(u_d, x_d, th), theta0, lr, max_steps=64)
        loop.run(n_it, graphed=(kind == "graph"))
        runs[kind] = loop.results()
        assert runs[kind]["iterations"] >= n_it
    g, e = runs["graph"], runs["eager"]
    assert np.array_equal(g["loss_trace"][:n_it], e["loss_trace"][:n_it]) and np.array_equal(g["parameter_trace"][:n_it], e["parameter_trace"][:n_it])
    theta, trace = theta0.copy(), []
    for k in range(n_it):
        loss, dp = sid.step([inputs[i] for i in range(inputs.shape[0])], [states[i] for i in range(states.shape[0])], theta)
        theta = theta - lr * dp
        trace.append((loss, theta.copy()))
    assert np.allclose(g["loss_trace"][:n_it], [a for a, _ in trace], rtol=1e-11, atol=0)
    assert np.abs(g["parameter_trace"][:n_it] - np.array([b for _, b in trace])).max() <= 1e-12 * max(1.0, np.abs(theta0).max())
    assert g["loss_trace"][n_it - 1] < g["loss_trace"][0]


def test_recmat_example_writes_the_reference_schema(tmp_path, golden_dir):
    """examples/oc_recmat_pdp.py (Examples/OC/quadrotor/uav_PDP_Recmat.py, rocket_PDP_Recmat.py:40-90): the loop runs, the loss falls, the .mat carries the reference's
    fields - and `true_solution`, OCSys.ocSolver from the driver's own initial state at its horizon, is the solution IPOPT stored for that problem
    (tests/golden/oc_quadrotor.npz: true_solution of Examples/OC/quadrotor/data/PDP_OC_results_trial_0.mat, cost 3122.99940085)"""
    import scipy.io as sio
    g = np.load(os.path.join(golden_dir, "oc_quadrotor.npz"))
    for extra in (["--graph"], []):
        out = str(tmp_path / ("r%d.mat" % len(extra)))
        txt = run("oc_recmat_pdp.py", "--system", "quadrotor", "--iters", "40", "--lr", "1e-4", "--sigma", "0.5", "--out", out, *extra)
        r = sio.loadmat(out)["results"][0, 0]
        assert set(["trail_no", "parameter_trace", "loss_trace", "learning_rate", "solved_solution", "true_solution", "time_passed", "dt", "horizon"]) <= set(r.dtype.names)
        L = r["loss_trace"].flatten()
        assert L.size == 40 and L[-1] < L[0], txt
        assert np.asarray(r["parameter_trace"]).shape == (41, 140) and int(r["horizon"].squeeze()) == 35
        ts, ss = r["true_solution"][0, 0], r["solved_solution"][0, 0]
        assert abs(float(ts["cost"].squeeze()) - float(g["true_cost"])) <= 1e-9 * float(g["true_cost"])
        assert np.abs(ts["state_traj_opt"] - g["true_state"]).max() <= 1e-6 and np.abs(ts["control_traj_opt"] - g["true_control"]).max() <= 1e-6
        assert ss["state_traj"].shape == (36, 13) and ss["control_traj"].shape == (35, 4)


def test_oc_example_stores_the_true_solution(tmp_path):
    import scipy.io as sio
    out = str(tmp_path / "oc.mat")
    run("oc_pdp.py", "--system", "quadrotor", "--horizon", "20", "--iters", "5", "--batch", "4", "--out", out)
    r = sio.loadmat(out)["results"][0, 0]
    ts = r["true_solution"][0, 0]
    assert ts["state_traj_opt"].shape == (21, 13) and ts["control_traj_opt"].shape == (20, 4) and ts["costate_traj_opt"].shape == (20, 13)
    assert float(ts["cost"].squeeze()) <= float(r["solved_solution"][0, 0]["cost"].squeeze()) * (1 + 1e-9)      # the optimum bounds any policy's cost from below
