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
