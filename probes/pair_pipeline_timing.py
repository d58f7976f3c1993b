"""ControlPlanning.step (Lagrange policy) and SysID.step: one-wave kernels against the rollout / sensitivity wave pairs (csrc/pdp_cp_pair_kernels.h) over horizons and
batch sizes.  Run once per variant: PDP_CP_POLY_VARIANT / PDP_SYSID_VARIANT = 1 (one wave) or 2 (pair, default)."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt
import bench
tag = "variant %s/%s" % (os.environ.get("PDP_CP_POLY_VARIANT", "2"), os.environ.get("PDP_SYSID_VARIANT", "2"))
rng = np.random.default_rng(0)
mdl = zoo.get("quadrotor", "sysid")
for B in (256, 1024):
    for T in (25, 50, 100, 200):
        u = rt.dev(0.3 * rng.standard_normal((B, T, mdl.m)))
        xobs = 0.3 * rng.standard_normal((B, T + 1, mdl.n)); xobs[:, :, 6] += 1.0
        xobs = rt.dev(xobs)
        th = rt.dev(1.0 + 0.2 * rng.uniform(-1, 1, mdl.p))
        ms = bench._event_ms(torch, lambda: mdl.sysid_step(u, xobs, th), reps=20, warm=3)
        print("%s sysid quadrotor B=%4d T=%3d: %.4f ms = %5.0f ns per step" % (tag, B, T, ms, ms * 1e6 / T), flush=True)
mdl = zoo.get("quadrotor", "oc")
for B in (256, 1024):
    for T in (25, 50, 100):
        npiv = 6
        pol = rt.make_policy("poly", pivots=np.linspace(0, T, npiv))
        x0 = 0.3 * rng.standard_normal((B, mdl.n)); x0[:, 6] = 1.0
        x0 = rt.dev(x0)
        th = rt.dev(0.3 * rng.standard_normal(npiv * mdl.m))
        ms = bench._event_ms(torch, lambda: mdl.cp_step(pol, npiv * mdl.m, x0, th, T), reps=20, warm=3)
        print("%s cp    quadrotor B=%4d T=%3d p=24: %.4f ms = %5.0f ns per step" % (tag, B, T, ms, ms * 1e6 / T), flush=True)
