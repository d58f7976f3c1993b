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
