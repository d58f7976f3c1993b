"""Timing of the multiple-shooting OC solver (product libraries) on bench.py's C2 / C3 workloads and the C4 shard: cold and warm solves.
Run once per kernel variant (PDP_MS_VARIANT=1: one wave per trajectory, 2: runner / evaluator pair) and compare; the solutions are written to
gpurun_out/ so that the two runs can be diffed."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt, JinEnv
import bench
variant = os.environ.get("PDP_MS_VARIANT", "2")
rng = np.random.default_rng(0)
out = {}
for system, B, T in (("cartpole", 256, 50), ("quadrotor", 1024, 50), ("rocket", 512, 100), ("quadrotor", 4096, 50), ("cartpole", 4096, 50)):
    mdl = zoo.get(system, "irl")
    if system == "cartpole":
        th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
        x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-0.5, 0.5, B)
        theta1 = th_star[None] + rng.uniform(-0.05, 0.05, (B, 7))
    elif system == "quadrotor":
        th_star = np.array(bench.THETA)
        x0 = bench.synth_inputs(B, 5)[0]
        theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR)))
    else:
        th_star = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
        x0 = np.zeros((B, 13)); x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3)); x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, 10)))
    x0d, theta1 = rt.dev(x0), rt.dev(theta1)
    demo = mdl.oc_solve_ms(x0d, th_star, T)
    cold_ms = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, th_star, T), reps=3, warm=1)
    warm = (demo["state"], demo["control"], demo["costate"])
    sol = mdl.oc_solve_ms(x0d, theta1, T, warm=warm, want_gains=True)
    warm_ms = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, theta1, T, warm=warm), reps=7, warm=2)
    itc, itw = demo["iterations"].double(), sol["iterations"].double()
    print("variant %s %-9s B=%4d T=%3d: cold %.3f ms (%d/%d converged, iterations mean %.2f max %d, status %s) | warm %.3f ms (%d/%d, iterations mean %.2f max %d, status %s)" %
          (variant, system, B, T, cold_ms, int(demo["converged"].sum()), B, itc.mean(), itc.max(), np.unique(demo["status"].cpu().numpy()),
           warm_ms, int(sol["converged"].sum()), B, itw.mean(), itw.max(), np.unique(sol["status"].cpu().numpy())), flush=True)
    key = "%s_%d" % (system, B)
    for k in ("state", "control", "costate", "cost"):
        out[key + "_cold_" + k] = demo[k].cpu().numpy()[:64]
        out[key + "_warm_" + k] = sol[k].cpu().numpy()[:64]
    out[key + "_warm_gains"] = sol["gains"].cpu().numpy()[:64]
    out[key + "_cold_iters"] = demo["iterations"].cpu().numpy()
    out[key + "_warm_iters"] = sol["iterations"].cpu().numpy()
os.makedirs("gpurun_out/r3b", exist_ok=True)
np.savez("gpurun_out/r3b/ms_variant_%s.npz" % variant, **out)
other = "gpurun_out/r3b/ms_variant_%s.npz" % ("1" if variant == "2" else "2")
if os.path.exists(other):
    o = np.load(other)
    for k in sorted(out):
        a, b = out[k], o[k]
        if a.dtype.kind == "f":
            print("  %-32s max |diff| vs the other variant: %.3e (scale %.3e)" % (k, np.abs(a - b).max(), np.abs(b).max()))
        else:
            print("  %-32s iterations differ in %d of %d trajectories" % (k, int((a != b).sum()), a.size))
