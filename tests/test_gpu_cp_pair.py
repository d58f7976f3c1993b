"""cp_step_poly2_kernel and sysid_step2_kernel (rollout wave + sensitivity wave per trajectory, csrc/pdp_cp_pair_kernels.h) against cp_step_poly_kernel /
sysid_step_kernel (one wavefront per trajectory):
the same arithmetic in the same order - loss, trajectory and gradient must agree BIT FOR BIT - over the workgroup shapes (1 / 2 / 4 trajectories per workgroup by
batch size), one to four parameter tiles, horizons of one chunk and of several, per-sample parameters, and a batch that is not a multiple of the workgroup.  The
one-wave kernel is pinned on the reference's own ControlPlanning.step runs (ref_cp_*_poly.npz, tests/test_gpu_models.py) and on the oracle at C3 / C4 sizes
(tests/test_gpu_configs.py - which run the pair kernel by default); this file transfers the pin between the two."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (system, pivots, horizon, batch): p = pivots x m -> 1 .. 4 tiles of 16 parameters; batches around the 256 / 512 CU thresholds of the workgroup shape
CASES = [("quadrotor", 6, 50, 1024), ("quadrotor", 6, 50, 1027), ("quadrotor", 3, 20, 5), ("quadrotor", 10, 37, 300), ("quadrotor", 16, 64, 131), ("rocket", 6, 100, 512),
         ("rocket", 6, 100, 700), ("cartpole", 5, 70, 9), ("pendulum", 12, 33, 260), ("robotarm", 7, 35, 64), ("quadrotor", 4, 1, 3), ("quadrotor", 4, 2, 3)]


# SysID.step: (system, horizon, batch)
SYSID_CASES = [("quadrotor", 100, 1024), ("quadrotor", 40, 1025), ("rocket", 70, 300), ("cartpole", 33, 7), ("pendulum", 1, 3), ("robotarm", 2, 70), ("quadrotor", 100, 513)]


def _run_all():
    sys.path.insert(0, ROOT)
    from pdp_amd import runtime as rt, zoo
    out = {}
    for ci, (system, T, B) in enumerate(SYSID_CASES):
        mdl = zoo.get(system, "sysid")
        rng = np.random.default_rng(500 + ci)
        u = 0.3 * rng.standard_normal((B, T, mdl.m))
        xobs = 0.3 * rng.standard_normal((B, T + 1, mdl.n))
        if system in ("quadrotor", "rocket"):
            xobs[:, :, 6] += 1.0
        th = 1.0 + 0.2 * rng.uniform(-1, 1, mdl.p)
        loss, grad = mdl.sysid_step(u, xobs, th)
        lossb, gradb = mdl.sysid_step(u, xobs, th[None] * (1 + 0.05 * rng.standard_normal((B, mdl.p))))
        for k, v in (("loss", loss), ("grad", grad), ("loss_b", lossb), ("grad_b", gradb)):
            out["sysid%d_%s" % (ci, k)] = v.cpu().numpy()
    for ci, (system, npiv, T, B) in enumerate(CASES):
        mdl = zoo.get(system, "oc")
        rng = np.random.default_rng(300 + ci)
        p = npiv * mdl.m
        x0 = 0.3 * rng.standard_normal((B, mdl.n))
        if system in ("quadrotor", "rocket"):
            x0[:, 6] = 1.0
        th = 0.3 * rng.standard_normal(p)
        pol = rt.make_policy("poly", pivots=np.linspace(0, T, npiv))
        loss, grad, x, u = mdl.cp_step(pol, p, x0, th, T, want_traj=True)
        thb = th[None] * (1 + 0.05 * rng.standard_normal((B, p)))            # per-sample parameters
        loss2, grad2 = mdl.cp_step(pol, p, x0, thb, T)
        for k, v in (("loss", loss), ("grad", grad), ("x", x), ("u", u), ("loss_b", loss2), ("grad_b", grad2)):
            out["%d_%s" % (ci, k)] = v.cpu().numpy()
    return out


def test_pair_kernel_equals_one_wave_kernel(tmp_path):
    """both sides in subprocesses: the kernel choice is read from the environment once per process (PDP_CP_POLY_VARIANT=3: the pair for every batch size)"""
    res = {}
    for tag, env in (("pair", dict(PDP_CP_POLY_VARIANT="3", PDP_SYSID_VARIANT="2")), ("onewave", dict(PDP_CP_POLY_VARIANT="1", PDP_SYSID_VARIANT="1"))):
        f = str(tmp_path / (tag + ".npz"))
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_cp_pair as m; np.savez(%r, **m._run_all())"
                % (ROOT, os.path.join(ROOT, "tests"), f))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        res[tag] = np.load(f)
    new, ref = res["pair"], res["onewave"]
    worst = {}
    for k in sorted(new.files):
        a, b = new[k], ref[k]
        assert np.isfinite(a).all(), k
        if not np.array_equal(a, b):
            worst[k] = float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
    assert not worst, "pair kernels differ from the one-wave kernels: %s" % json.dumps(worst)
