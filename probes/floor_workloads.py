"""One BASELINE kernel, launched a few times and nothing else: the workload of the SQ counter passes behind `floor_frac` (probes/profile_r06.sh, round-5 verdict items 5 and 7).
    python probes/floor_workloads.py <name> [launches]
names: sysid (C5a: SysID.step, quadrotor T=100 p=5 B=1024) | cp_poly (C3: ControlPlanning.step, Lagrange policy, quadrotor T=50 p=24 B=1024) | cp_poly_c4 (rocket T=100 p=18 B=512)
       | mlp (C5b: tanh-MLP [13,13], p=420, T=100, B=1024) | oc_c4 (fused OC unit, rocket T=100 p=10, B=512: a GPU's shard of C4, TPW = 2) | headline (C3 fused OC unit B=1024)
       | solve (C3 OC solve, plain warm start, B=1024) | solve_c2 (cart-pole, B=256)
Inputs are bench.py's (same seeds, same shapes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402
from pdp_amd import JinEnv, runtime as rt, zoo  # noqa: E402


def make(name):
    rng = np.random.default_rng(0)
    if name == "sysid":
        mdl = zoo.get("quadrotor", "sysid")
        B, T = 1024, 100
        u5 = rt.dev(rng.uniform(-1, 1, (B, T, 4)) + 2.5)
        x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
        xobs = mdl.sysid_integrate(x0, u5, np.array([1, 1, 1, 1, .4]))
        th5 = rt.dev(np.array([1.1, .95, 1.08, 1.03, .38]))
        return lambda: mdl.sysid_step(u5, xobs, th5)
    if name in ("cp_poly", "cp_poly_c4"):
        system, B, T, p = ("quadrotor", 1024, 50, 24) if name == "cp_poly" else ("rocket", 512, 100, 18)
        mdl = zoo.get(system, "oc")
        x0 = np.zeros((B, 13))
        if system == "quadrotor":
            x0[:, :3] = rng.uniform(-5, 5, (B, 3))
            x0[:, 6] = 1
        else:
            x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
            x0[:, 3] = -0.1
            x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        x0d, thp, pol = rt.dev(x0), rt.dev(0.5 * rng.standard_normal(p)), rt.make_policy("poly", pivots=np.linspace(0, T, 6))
        return lambda: mdl.cp_step(pol, p, x0d, thp, T)
    if name == "mlp":
        mdl = zoo.get("quadrotor", "oc")
        B, T, p = 1024, 100, 420
        x0 = np.zeros((B, 13))
        x0[:, :3] = rng.uniform(-2, 2, (B, 3))
        x0[:, 6] = 1
        x0d, thp, pol = rt.dev(x0), rt.dev(0.1 * rng.standard_normal(p)), rt.make_policy("mlp", layers=[13, 13, 4])
        return lambda: mdl.cp_step(pol, p, x0d, thp, T)
    if name == "oc_c4":
        mdl = zoo.get("rocket", "irl")
        B, T = 512, 100
        th4 = rt.dev(np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]))
        x0 = np.zeros((B, 13))
        x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
        x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        x0d = rt.dev(x0)
        u4 = rt.dev(np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3)))
        dx4, du4 = rt.dev(np.zeros((B, T + 1, 13))), rt.dev(np.zeros((B, T, 3)))
        bufs = {}
        return lambda: mdl.oc_pdp_grad(u4, th4, dx4, du4, x0=x0d, buffers=bufs)
    if name == "headline":
        mdl = zoo.get("quadrotor", "irl")
        x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(1024, 1000))
        th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
        bufs = {}
        return lambda: mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs, packed=True)
    if name in ("solve", "solve_c2"):
        system, B, T = ("quadrotor", 1024, 50) if name == "solve" else ("cartpole", 256, 50)
        mdl = zoo.get(system, "irl")
        if system == "cartpole":
            th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
            x0 = np.zeros((B, 4))
            x0[:, 1] = rng.uniform(-0.5, 0.5, B)
            theta1 = th_star[None] + rng.uniform(-0.05, 0.05, (B, 7))
        else:
            th_star = np.array(bench.THETA)
            x0 = bench.synth_inputs(B, 5)[0]
            theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR)))
        x0d, theta1 = rt.dev(x0), rt.dev(theta1)
        demo = mdl.oc_solve_ms(x0d, th_star, T)
        warm = (demo["state"], demo["control"], demo["costate"])
        return lambda: mdl.oc_solve_ms(x0d, theta1, T, warm=warm)
    raise SystemExit("unknown workload " + name)


if __name__ == "__main__":
    fn = make(sys.argv[1])
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        fn()
    torch.cuda.synchronize()
