"""cp_step_mlp16_kernel (network in registers, csrc/pdp_cp_mlp_kernels.h) against the general adjoint kernel (cp_step_adjoint_kernel): the same arithmetic in
the same order: loss and trajectory must agree BIT FOR BIT and the gradient to the last bits (<= 1e-15 of its largest entry: the compiler contracts a few
products of the adjoint sweep differently in the two kernels), for every network shape the register kernel accepts (1-4 layers, widths <= 16) and
for a shape it must hand back to the general kernel.  The general kernel is pinned on the reference's own ControlPlanning.step runs (ref_cp_*_mlp.npz,
tests/test_gpu_models.py); this file transfers that pin."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [("quadrotor", [13, 13, 4], 100, 6), ("quadrotor", [13, 4], 37, 3), ("quadrotor", [4], 20, 2), ("quadrotor", [8, 5, 7, 4], 50, 3),
         ("quadrotor", [16, 16, 4], 64, 2), ("cartpole", [4, 4, 1], 60, 5), ("pendulum", [2, 1], 30, 4), ("robotarm", [4, 4, 2], 35, 4), ("rocket", [13, 13, 3], 100, 3),
         ("quadrotor", [20, 4], 15, 2)]       # the last one is beyond the register kernel: both variants run the general kernel


def _run_all():
    sys.path.insert(0, ROOT)
    import torch
    from pdp_amd import runtime as rt, zoo
    out = {}
    for ci, (system, layers, T, B) in enumerate(CASES):
        mdl = zoo.get(system, "oc")
        rng = np.random.default_rng(100 + ci)
        sizes = [mdl.n] + layers
        p = sum(sizes[k + 1] * sizes[k] + sizes[k + 1] for k in range(len(layers)))
        x0 = 0.3 * rng.standard_normal((B, mdl.n))
        if system in ("quadrotor", "rocket"):
            x0[:, 6] = 1.0
        th = 0.2 * rng.standard_normal(p)
        pol = rt.make_policy("mlp", layers=layers)
        loss, grad, x, u = mdl.cp_step(pol, p, x0, th, T, want_traj=True)
        thb = th[None] * (1 + 0.05 * rng.standard_normal((B, p)))            # per-sample parameters
        loss2, grad2 = mdl.cp_step(pol, p, x0, thb, T)
        for k, v in (("loss", loss), ("grad", grad), ("x", x), ("u", u), ("loss_b", loss2), ("grad_b", grad2)):
            out["%d_%s" % (ci, k)] = v.cpu().numpy()
    return out


def test_register_kernel_equals_general_kernel(tmp_path):
    new = _run_all()
    ref_file = str(tmp_path / "general.npz")
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_cp_mlp as m; np.savez(%r, **m._run_all())"
            % (ROOT, os.path.join(ROOT, "tests"), ref_file))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PDP_CP_MLP_VARIANT="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    ref = np.load(ref_file)
    worst = {}
    for k in sorted(new):
        a, b = new[k], ref[k]
        assert np.isfinite(a).all(), k
        err = float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
        if err > (1e-15 if "grad" in k else 0.0):
            worst[k] = err
    assert not worst, "register kernel differs from the general kernel: %s" % json.dumps(worst)
