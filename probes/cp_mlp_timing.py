"""C5b (quadrotor neural-policy ControlPlanning.step, hidden [13, 13], p = 420, T = 100): kernel time of the variant selected by PDP_CP_MLP_VARIANT over batch sizes."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt
import bench
mdl = zoo.get("quadrotor", "oc")
rng = np.random.default_rng(0)
T, p = 100, 420
pol = rt.make_policy("mlp", layers=[13, 13, 4])
thp = rt.dev(0.1 * rng.standard_normal(p))
for B in (256, 1024, 2048, 8192):
    x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1
    x0d = rt.dev(x0)
    ms = bench._event_ms(torch, lambda: mdl.cp_step(pol, p, x0d, thp, T), reps=5, warm=1)
    print("variant %s B=%5d: %.3f ms = %.2f M trajectories/s = %d cycles per time step and wave at 2.1 GHz" % (os.environ.get("PDP_CP_MLP_VARIANT", "2"), B, ms, B / ms / 1e3, ms * 1e-3 * 2.1e9 / T / max(1, B / 1024)))
