"""Timing of the packed small-system LQR kernel (four trajectories per wavefront) on cart-pole sizes: n=4 m=1 p=7 T=50."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pdp_amd import runtime as rt
rng = np.random.default_rng(0)
n, m, T = 4, 1, 50
for p in (7, 1):
    for B in (256, 1024, 4096, 16384):
        F = rt.dev(np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n))); G = rt.dev(0.3 * rng.standard_normal((B, T, n, m)))
        E = rt.dev(0.1 * rng.standard_normal((B, T, n, p))); Hxx = rt.dev(np.tile(np.eye(n), (B, T, 1, 1))); Huu = rt.dev(np.tile(0.5 * np.eye(m), (B, T, 1, 1)))
        Hxe, Hue = rt.dev(0.2 * rng.standard_normal((B, T, n, p))), rt.dev(0.2 * rng.standard_normal((B, T, m, p)))
        hxx, hxe = rt.dev(np.tile(np.eye(n), (B, 1, 1))), rt.dev(0.2 * rng.standard_normal((B, n, p)))
        for wc in (True, False):
            fn = lambda: rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxe=Hxe, Hue=Hue, want_costate=wc)
            for _ in range(3): fn()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); fn(); b.record()
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            print("cart-pole LQR n=4 m=1 p=%d T=50 B=%5d costate=%d: %.3f ms  %.2f M traj/s" % (p, B, wc, ms, B / ms / 1e3))
