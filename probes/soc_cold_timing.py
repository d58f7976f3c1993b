"""Cold solves (zero guess) of bench.py's C2 / C3 batches with and without the second-order correction (PDP_MS_WITH_SOC): time, iterations, corrected steps, and whether the
two land in the same optima.  A correction costs a full Newton sweep in this kernel (IPOPT: a back-substitution).
    python probes/soc_cold_timing.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                # noqa: E402
from pdp_amd import runtime as rt, zoo                      # noqa: E402

torch = rt.torch_cuda()
rng = np.random.default_rng(0)
for system, B, T in (("cartpole", 256, 50), ("cartpole", 1024, 50), ("quadrotor", 1024, 50), ("rocket", 512, 40), ("robotarm", 512, 35)):
    mdl = zoo.get(system, "irl")
    if system == "cartpole":
        th = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
        x0 = np.zeros((B, 4))
        x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    elif system == "quadrotor":
        th = np.array(bench.THETA)
        x0 = bench.synth_inputs(B, 5)[0]
    else:
        d = np.load(os.path.join(ROOT, "tests", "golden", "demos_%s.npz" % system))
        th = d["true_parameter"]
        x0 = d["state"][0, 0][None] * (1 + 0.2 * rng.uniform(-1, 1, (B, d["state"].shape[2])))
    x0d = rt.dev(x0)
    res = {}
    for soc in (False, True):
        ms = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, th, T, soc=soc), reps=3, warm=1)
        s = mdl.oc_solve_ms(x0d, th, T, soc=soc)
        res[soc] = s
        it = s["iterations"].float()
        print("%-9s B %4d T %3d  soc %-5s  %.3f ms  converged %4d / %d  iterations mean %.2f max %d  corrected-step trajectories %d  other status bits %d" % (
            system, B, T, soc, float(ms), int(s["converged"].sum()), B, float(it.mean()), int(it.max()), int(((s["status"] & 1024) != 0).sum()),
            int(((s["status"] & ~(1024 | 128)) != 0).sum())))
    both = res[False]["converged"] & res[True]["converged"]
    dc = ((res[False]["cost"] - res[True]["cost"]).abs() / res[False]["cost"].abs().clamp(min=1.0))[both]
    print("          same optimum (cost within 1e-8) on %d of the %d trajectories both converged on; largest relative cost difference %.2e" % (int((dc <= 1e-8).sum()), int(both.sum()), float(dc.max()) if dc.numel() else 0.0))
