#!/usr/bin/env python3
"""C5b: ControlPlanning.step with the tanh MLP [13, 13] (p = 420), T = 100 - the four-trajectory MFMA kernel (round 5, default) against the one-trajectory register
kernel (PDP_CP_MLP_VARIANT=3, round 4's default) over batch sizes.  HIP-event medians."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pdp_amd import runtime as rt, zoo
mdl = zoo.get("quadrotor", "oc")
rng = np.random.default_rng(0)
T, p = 100, 420
pol = rt.make_policy("mlp", layers=[13, 13, 4])
thp = rt.dev(0.1 * rng.standard_normal(p))
print("PDP_CP_MLP_VARIANT =", os.environ.get("PDP_CP_MLP_VARIANT", "2 (default)"))
for B in (64, 256, 512, 1024, 2048, 4096, 8192, 16384):
    x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1
    x0d = rt.dev(x0)
    ms = bench._event_ms(torch, lambda: mdl.cp_step(pol, p, x0d, thp, T), reps=7, warm=2)
    print("B = %6d   %.4f ms   %.2f M trajectories/s   %.0f ns per step and round of 1024" % (B, ms, B / ms / 1e3, ms * 1e6 / T / max(1, B / 1024)))
