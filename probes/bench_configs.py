"""Throughput of the other BASELINE.json configurations on one MI355X (context for DESIGN.md; the headline line is bench.py)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from pdp_amd import zoo, runtime as rt, JinEnv, ocsolver

def timeit(fn, n=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

res = {}
rng = np.random.default_rng(0)
# C4: rocket planning T=100, p=18, B=512 per GPU
mdl = zoo.get("rocket", "oc"); B, T, p = 512, 100, 18
x0 = np.zeros((B, 13)); x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3)); x0[:, 3] = -.1; x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
x0d, th = rt.dev(x0), rt.dev(0.5 * rng.standard_normal(p)); pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
dt = timeit(lambda: mdl.cp_step(pol, p, x0d, th, T)); res["C4 rocket ControlPlanning.step T=100 p=18 B=512"] = (B / dt, dt * 1e3)
# C3 U-CP: quadrotor planning T=50 p=24 B=1024
mdl = zoo.get("quadrotor", "oc"); B, T, p = 1024, 50, 24
x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-5, 5, (B, 3)); x0[:, 6] = 1
x0d, th = rt.dev(x0), rt.dev(rng.standard_normal(p)); pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
dt = timeit(lambda: mdl.cp_step(pol, p, x0d, th, T)); res["C3 quadrotor ControlPlanning.step T=50 p=24 B=1024"] = (B / dt, dt * 1e3)
# C5a: quadrotor SysID T=100 p=5 B=1024
mdl = zoo.get("quadrotor", "sysid"); B, T = 1024, 100
u = rt.dev(rng.uniform(-1, 1, (B, T, 4)) + 2.5); x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
xobs = mdl.sysid_integrate(x0, u, np.array([1, 1, 1, 1, .4])); th = rt.dev(np.array([1.1, .95, 1.08, 1.03, .38]))
dt = timeit(lambda: mdl.sysid_step(u, xobs, th)); res["C5a quadrotor SysID.step T=100 p=5 B=1024"] = (B / dt, dt * 1e3)
# C5b: quadrotor neural policy [13,13] p=420 T=100 (composed from modular kernels) B=256
mdl = zoo.get("quadrotor", "oc"); B, T, p = 256, 100, 420
x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1
x0d, th = rt.dev(x0), rt.dev(0.1 * rng.standard_normal(p)); pol = rt.make_policy("mlp", layers=[13, 13, 4])
dt = timeit(lambda: mdl.cp_step(pol, p, x0d, th, T), n=3, warm=1); res["C5b quadrotor MLP-policy step T=100 p=420 B=256 (fused adjoint kernel)"] = (B / dt, dt * 1e3)
B = 1024; x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1; x0d = rt.dev(x0)
dt = timeit(lambda: mdl.cp_step(pol, p, x0d, th, T), n=3, warm=1); res["C5b quadrotor MLP-policy step T=100 p=420 B=1024 (one GPU's shard of C5)"] = (B / dt, dt * 1e3)
# C2: cart-pole IRL iteration (OC solve warm-started + PDP gradient), T=50, B=256 per-sample theta
from test_gpu_ocsolver import make_oc
oc = make_oc("cartpole"); B, T = 256, 50
x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-.5, .5, B); th_star = np.array([.5, .5, 1, 1, 6, 1, 1.])
demo = ocsolver.solve_batch(oc, x0, T, th_star, want_gains=True)
theta = th_star[None] + rng.uniform(-.05, .05, (B, 7))
t0 = time.perf_counter(); sol = ocsolver.solve_batch(oc, x0, T, theta, warm_start=demo, want_gains=True); torch.cuda.synchronize(); t_solve = time.perf_counter() - t0
dt = timeit(lambda: oc.pdp_grad_batch(sol["control"], theta, demo["state"], demo["control"], state_traj=sol["state"], costate_traj=sol["costate"]))
res["C2 cart-pole aux+Riccati+grad (given optimum) T=50 p=7 B=256"] = (B / dt, dt * 1e3)
res["C2 cart-pole OC solve, closed-loop warm start, %d iterations, B=256" % sol["iterations"]] = (B / t_solve, t_solve * 1e3)
# materialised API route of the reference on C3 sizes: getAuxSys (dense matrices to HBM) then lqrSolver (dense matrices from HBM)
mdl = zoo.get("quadrotor", "irl"); B, T = 1024, 50
import bench as _b
x0, u, dx, du = (rt.dev(a) for a in _b.synth_inputs(B, 7)); th = rt.dev(np.array(_b.THETA))
x, _ = mdl.oc_rollout(x0, u, th); lam = mdl.oc_costate(x, u, th)
dt = timeit(lambda: mdl.oc_auxsys(x, u, lam, th)); res["C3 materialised OCSys.getAuxSys (all 11 families to HBM), quadrotor T=50 B=1024"] = (B / dt, dt * 1e3)
aux = mdl.oc_auxsys(x, u, lam, th)
dt = timeit(lambda: rt.lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], aux["hxe"], E=aux["dynE"], Hxu=aux["Hxu"], Hxe=aux["Hxe"], Hue=aux["Hue"]))
res["C3 materialised LQR.lqrSolver (X,U,Lambda out), quadrotor T=50 p=9 B=1024"] = (B / dt, dt * 1e3)
# materialised ControlPlanning.integrateAuxSys / SysID.integrateAuxSys (random operands of the C3 / C5a shapes: timing only)
B, T, n, m, p = 1024, 50, 13, 4, 24
Fm, Gm, Uxm, Uem = (rt.dev(rng.standard_normal(s) * .1) for s in ((B, T, n, n), (B, T, n, m), (B, T, m, n), (B, T, m, p)))
dt = timeit(lambda: rt.cp_aux_integrate(Fm, Gm, Uxm, Uem)); res["C3 materialised ControlPlanning.integrateAuxSys, n=13 m=4 p=24 T=50 B=1024"] = (B / dt, dt * 1e3)
B, T, n, p = 1024, 100, 13, 5
Fm, Em = (rt.dev(rng.standard_normal(s) * .1) for s in ((B, T, n, n), (B, T, n, p)))
dt = timeit(lambda: rt.sysid_aux_integrate(Fm, Em)); res["C5a materialised SysID.integrateAuxSys, n=13 p=5 T=100 B=1024"] = (B / dt, dt * 1e3)
for k, (v, ms) in res.items(): print("%-82s %12.0f traj/s  %9.3f ms" % (k, v, ms))
json.dump({k: {"traj_per_s": v, "ms": ms} for k, (v, ms) in res.items()}, open("gpurun_out/bench_configs.json", "w"), indent=1)
