"""The headline kernel (oc_pdp_fused3_kernel) on batches that do not fill the chip: trajectories per workgroup chosen by the library (1 / 2 / 4 by batch size) against
the fixed 4 of round 2 (PDP_FUSED_TPW=4).  C3 (quadrotor T = 50) and the C4 shard (rocket T = 100, B = 512).  Run once per setting."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt, JinEnv
import bench
tag = "TPW=" + os.environ.get("PDP_FUSED_TPW", "auto")
mdl = zoo.get("quadrotor", "irl")
th = rt.dev(np.array(bench.THETA))
for B in (128, 256, 512, 1024):
    x0, u, dx, du = (rt.dev(a) for a in bench.synth_inputs(B, 1000))
    bufs = {}
    ms = bench._event_ms(torch, lambda: mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs), reps=20, warm=3)
    print("%-9s quadrotor T=50  B=%4d: %.4f ms = %.2f M trajectories/s" % (tag, B, ms, B / ms / 1e3))
mdl = zoo.get("rocket", "irl")
rng = np.random.default_rng(0)
T = 100
th4 = rt.dev(np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]))
for B in (256, 512, 1024):
    x0 = np.zeros((B, 13)); x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3)); x0[:, 3] = -0.1; x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
    u4 = rt.dev(np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3)))
    x0d, dx4, du4 = rt.dev(x0), rt.dev(np.zeros((B, T + 1, 13))), rt.dev(np.zeros((B, T, 3)))
    bufs = {}
    ms = bench._event_ms(torch, lambda: mdl.oc_pdp_grad(u4, th4, dx4, du4, x0=x0d, buffers=bufs), reps=20, warm=3)
    print("%-9s rocket    T=100 B=%4d: %.4f ms = %.2f M trajectories/s" % (tag, B, ms, B / ms / 1e3))
