#!/usr/bin/env python3
"""Rollout pre-pass of ControlPlanning.step with the Lagrange policy (round 4; C4's ControlPlanning half: rocket, T = 100, p = 18): batches with several trajectories per SIMD
roll out beforehand with ONE LANE per trajectory (cp_poly_rollout_lanes_kernel) and run the sensitivity kernel on the given trajectories (cp_step_poly_kernel<.., GIVEN>: only the
Jacobian pool and the basis in LDS, PDP_CP_GIVEN_WGS workgroups per CU) instead of the rollout wave + sensitivity wave pair.  PDP_CP_PREPASS = 0 / 1 forces the mode (one
subprocess per setting); prints the time of ControlPlanning.step per batch size and the deviation of loss / gradient / trajectory from the pair kernel."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from pdp_amd import JinEnv, zoo, runtime as rt
mdl = zoo.get("rocket", "oc")
rng = np.random.default_rng(0)
T = 100
out = {}
pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
for B in (1024, 2048, 4096, 8192):
    x0 = np.zeros((B, 13))
    x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
    x0d = rt.dev(x0)
    th = rt.dev(0.5 * rng.standard_normal((B, 18)))          # per-sample parameters (the harder case for the pre-pass: every lane its own row)
    ms = bench._event_ms(torch, lambda: mdl.cp_step(pol, 18, x0d, th, T), reps=7, warm=2)
    l, g, x, u = mdl.cp_step(pol, 18, x0d, th, T, want_traj=True)
    out[str(B)] = [ms, l.cpu().numpy().tolist(), g.cpu().numpy().tolist(), x[::97].cpu().numpy().tolist(), u[::97].cpu().numpy().tolist()]
print("RESULT " + json.dumps(out))
""" % ROOT

if __name__ == "__main__":
    import json
    import numpy as np
    res = {}
    for mode in ("0", "1", "default"):
        env = dict(os.environ)
        env.pop("PDP_CP_PREPASS", None)
        if mode != "default":
            env["PDP_CP_PREPASS"] = mode
        r = subprocess.run([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(mode, "FAILED", r.stdout[-1500:])
            continue
        res[mode] = json.loads(line[0][7:])
    ref = res.get("0")
    for mode, d in res.items():
        dev = ""
        if ref:
            rel = lambda k: max(np.abs(np.array(d[B][k]) - np.array(ref[B][k])).max() / np.abs(np.array(ref[B][k])).max() for B in d)
            dev = "   largest deviation from the pair kernel: loss %.1e, gradient %.1e, states %.1e, controls %.1e (relative to the largest entry)" % (rel(1), rel(2), rel(3), rel(4))
        print("pre-pass %-8s " % mode + "  ".join("B=%s %.4f ms" % (B, v[0]) for B, v in d.items()) + dev)
