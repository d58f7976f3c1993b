"""cp_step_mlp4t_kernel / cp_step_mlp16_kernel (network in registers, csrc/pdp_cp_mlp_kernels.h) against the general adjoint kernel (cp_step_adjoint_kernel): the same arithmetic in
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


def _variant(tmp_path, v):
    ref_file = str(tmp_path / ("variant%s.npz" % v))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_gpu_cp_mlp as m; np.savez(%r, **m._run_all())"
            % (ROOT, os.path.join(ROOT, "tests"), ref_file))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PDP_CP_MLP_VARIANT=str(v)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return np.load(ref_file)


def _compare(new, ref, what):
    worst = {}
    for k in sorted(new):
        a, b = new[k], ref[k]
        assert np.isfinite(a).all(), k
        err = float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
        if err > (1e-15 if "grad" in k else 0.0):
            worst[k] = err
    assert not worst, "%s: %s" % (what, json.dumps(worst))


def test_register_kernels_equal_the_general_kernel(tmp_path):
    """default route (round 5: shared parameters -> cp_step_mlp4t_kernel, four trajectories per wavefront on the 4-block MFMA; per-sample parameters -> the one-trajectory
    register kernel) == PDP_CP_MLP_VARIANT=3 (the one-trajectory register kernel for everything: round 4's route) == PDP_CP_MLP_VARIANT=1 (the general kernel):
    loss, states and controls bit for bit, gradient to 1e-15 of its largest entry"""
    new = _variant(tmp_path, 4)            # (four per wavefront for every batch size: the default takes it from two trajectories per CU on)
    one = _variant(tmp_path, 3)
    gen = _variant(tmp_path, 1)
    _compare(one, gen, "register kernel (one trajectory per wavefront) differs from the general kernel")
    _compare(new, one, "four-trajectory MFMA kernel differs from the one-trajectory register kernel")


@pytest.mark.parametrize("B", [513, 514, 515, 517, 1024, 1027])
def test_four_trajectory_kernel_on_batches_that_do_not_fill_its_wavefronts(B):
    """cp_step_mlp4t_kernel packs trajectories 4 w .. 4 w + 3 into wavefront w: a batch that is no multiple of four (padding lanes repeat the last trajectory, their
    stores are dropped), with and without the trajectory outputs - every trajectory equals the same trajectory run in a batch of one (which takes the one-trajectory
    register kernel: the two kernels agree bit for bit in loss and trajectories, to the last bits in the gradient)"""
    sys.path.insert(0, ROOT)
    from pdp_amd import runtime as rt, zoo
    mdl = zoo.get("quadrotor", "oc")
    rng = np.random.default_rng(B)
    T, layers = 30, [13, 13, 4]
    p = 13 * 13 + 13 + 13 * 13 + 13 + 4 * 13 + 4
    x0 = 0.3 * rng.standard_normal((B, 13))
    x0[:, 6] = 1.0
    th = 0.2 * rng.standard_normal(p)
    pol = rt.make_policy("mlp", layers=layers)
    loss, grad, x, u = (a.cpu().numpy() for a in mdl.cp_step(pol, p, x0, th, T, want_traj=True))
    l2, g2 = (a.cpu().numpy() for a in mdl.cp_step(pol, p, x0, th, T))
    assert np.array_equal(l2, loss) and np.array_equal(g2, grad)
    for i in sorted(set([0, B // 2, B - 1])):
        l1, g1, x1, u1 = (a.cpu().numpy() for a in mdl.cp_step(pol, p, x0[i:i + 1], th, T, want_traj=True))
        assert np.array_equal(l1[0], loss[i]) and np.array_equal(x1[0], x[i]) and np.array_equal(u1[0], u[i]), i
        assert np.abs(g1[0] - grad[i]).max() <= 1e-15 * np.abs(grad[i]).max(), i
