"""The runner / evaluator kernel (oc_pdp_fused3_kernel, the default for n > 4) and the one-wave kernel (oc_pdp_fused_kernel, PDP_FUSED_VARIANT=1)
run the same generated code and the same MFMA sequences: their outputs must agree to rounding (1e-11 of the largest entry, 1e-9 over 260 steps; most arrays agree to the bit), for every horizon that exercises the chunk
schedule (single steps, remainders of the 4-step groups, one / two / three chunks, the longest horizon the two-buffer layout takes and the first
one it hands to the one-wave kernel), with and without a given trajectory, with the sensitivities written out."""
import os, subprocess, sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(%(here)r))
from pdp_amd import zoo
out = {}
cases = (("quadrotor", [1.0, 1.0, 1.0, 1.0, 0.4, 1.0, 1.0, 5.0, 1.0], (1, 2, 3, 5, 17, 18, 35, 50, 100, 259, 260)),
         ("rocket", [0.5, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 50.0, 1.0, 1.0], (4, 50, 100)))
for system, theta, horizons in cases:
    mdl = zoo.get(system, "irl")
    n, m, p = mdl.n, mdl.m, mdl.p
    assert p == len(theta)
    theta = np.asarray(theta, dtype=np.float64)
    rng = np.random.default_rng(7)
    for T in horizons:
        B = 9                                                    # not a multiple of 4: the last workgroup has idle wave pairs
        s = 1.0 if T <= 100 else 0.01        # long horizons: stay close to hover (an open-loop tumble amplifies the last bit over 260 steps)
        x0 = np.zeros((B, n))
        x0[:, 0:3] = rng.uniform(-2, 2, (B, 3))
        x0[:, 3:6] = 0.1 * s * rng.standard_normal((B, 3))
        q = np.concatenate([np.ones((B, 1)), 0.1 * s * rng.standard_normal((B, 3))], axis=1)
        x0[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
        x0[:, 10:13] = 0.05 * s * rng.standard_normal((B, 3))
        if system == "quadrotor":
            u = 2.5 + 0.05 * s * rng.standard_normal((B, T, m))     # hover thrust per rotor
        else:
            u = 0.05 * s * rng.standard_normal((B, T, m)); u[:, :, 0] += 10.0      # hover thrust along the body axis
        dx = 0.1 * rng.standard_normal((B, T + 1, n))
        du = 0.1 * rng.standard_normal((B, T, m))
        o = mdl.oc_pdp_grad(u, theta, dx, du, x0=x0, want_sens=True)
        for k in ("loss", "grad", "x", "lam", "dxdp", "dudp", "status"):
            out["%%s_T%%d_%%s" %% (system, T, k)] = o[k].cpu().numpy()
        og = mdl.oc_pdp_grad(u, theta, dx, du, x=o["x"].clone(), lam=o["lam"].clone(), packed=True)
        out["%%s_T%%d_given_packed" %% (system, T)] = og["packed"].cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def _run(variant, path):
    env = dict(os.environ, PDP_FUSED_VARIANT=str(variant))
    r = subprocess.run([sys.executable, "-c", WORKER % dict(here=HERE), path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    return np.load(path)


def test_runner_evaluator_kernel_is_bit_identical_to_the_one_wave_kernel(tmp_path):
    a = _run(3, str(tmp_path / "v3.npz"))
    b = _run(1, str(tmp_path / "v1.npz"))
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 80
    worst = {}
    for k in a.files:
        x, y = a[k], b[k]
        assert x.shape == y.shape, k
        assert np.isfinite(x.astype(np.float64)).all(), k
        if k.endswith("_status"):
            assert np.array_equal(x, y) and not x.any(), k
            continue
        # the same arithmetic, but two compilations of the generated code (the compiler contracts multiply-adds on its own in each) and different
        # chunking of the reductions: equal to rounding, not to the bit
        err = np.max(np.abs(x - y)) / max(1e-300, np.max(np.abs(y)))
        worst[k] = err
        T = int(k.split("_T")[1].split("_")[0])
        assert err < (1e-11 if T <= 100 else 1e-9), "%s: relative deviation %g" % (k, err)
    print("largest deviations:", sorted(worst.items(), key=lambda kv: -kv[1])[:5])
