"""How the materialised LQR kernel's time scales with batch and outputs (latency- or throughput-bound?)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt
import bench as _b
def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
mdl = zoo.get("quadrotor", "irl")
for B in (128, 256, 512, 1024, 2048, 4096):
    x0, u, dx, du = (rt.dev(a) for a in _b.synth_inputs(B, 7)); th = rt.dev(np.array(_b.THETA))
    x, _ = mdl.oc_rollout(x0, u, th); lam = mdl.oc_costate(x, u, th)
    aux = mdl.oc_auxsys(x, u, lam, th)
    for wc in (True, False):
        dt = timeit(lambda: rt.lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], aux["hxe"], E=aux["dynE"], Hxu=aux["Hxu"],
                                         Hxe=aux["Hxe"], Hue=aux["Hue"], want_costate=wc))
        print("B=%5d want_costate=%d  %.3f ms  %.2f M traj/s" % (B, wc, dt * 1e3, B / dt / 1e6))
