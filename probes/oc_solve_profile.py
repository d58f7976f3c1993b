"""Kernel-time breakdown of the C2 OC solve (run under rocprofv3 --kernel-trace --stats)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from pdp_amd import ocsolver
from test_gpu_ocsolver import make_oc
rng = np.random.default_rng(0)
oc = make_oc("cartpole"); B, T = 256, 50
x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-.5, .5, B); th_star = np.array([.5, .5, 1, 1, 6, 1, 1.])
demo = ocsolver.solve_batch(oc, x0, T, th_star, want_gains=True)
theta = th_star[None] + rng.uniform(-.05, .05, (B, 7))
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sol = ocsolver.solve_batch(oc, x0, T, theta, warm_start=demo, want_gains=True)
    torch.cuda.synchronize(); print("warm solve %.2f ms, iterations %d, converged %d" % ((time.perf_counter() - t0) * 1e3, sol["iterations"], int(sol["converged"].sum())))
