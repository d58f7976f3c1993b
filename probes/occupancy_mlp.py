#!/usr/bin/env python3
"""Occupancy mode of cp_step_mlp16_kernel (round 4): ControlPlanning.step with the tanh MLP [13, 13] (p = 420), quadrotor, T = 100 (C5b).  PDP_CP_MLP_LDS_KB = 40: pool rows
sized for four workgroups per CU (one wavefront per SIMD, rounds 2 - 3); 20: eight per CU (two per SIMD; the kernel's 240 VGPRs allow it since its pool rows carry their own
zero and constants); default: the library's rule (20 for batches beyond one trajectory per SIMD).  One subprocess per setting."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from pdp_amd import zoo, runtime as rt
mdl = zoo.get("quadrotor", "oc")
rng = np.random.default_rng(0)
T = 100
pol = rt.make_policy("mlp", layers=[13, 13, 4])
thp = rt.dev(0.1 * rng.standard_normal(420))
out = {}
for B in (1024, 2048, 4096, 8192):
    x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1
    x0d = rt.dev(x0)
    ms = bench._event_ms(torch, lambda: mdl.cp_step(pol, 420, x0d, thp, T), reps=5, warm=2)
    l, g = mdl.cp_step(pol, 420, x0d, thp, T)
    out[str(B)] = [ms, l.cpu().numpy().tolist(), g[::101].cpu().numpy().tolist()]
print("RESULT " + json.dumps(out))
""" % ROOT

if __name__ == "__main__":
    import json
    import numpy as np
    res = {}
    for kb in ("40", "20", "13", "default"):
        env = dict(os.environ)
        env.pop("PDP_CP_MLP_LDS_KB", None)
        if kb != "default":
            env["PDP_CP_MLP_LDS_KB"] = kb
        r = subprocess.run([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(kb, "FAILED", r.stdout[-1500:])
            continue
        res[kb] = json.loads(line[0][7:])
    ref = res.get("40")
    for kb, d in res.items():
        dev = ""
        if ref:
            rel = lambda k: max(np.abs(np.array(d[B][k]) - np.array(ref[B][k])).max() / np.abs(np.array(ref[B][k])).max() for B in d)
            dev = "   largest deviation from the 40 KB layout: loss %.1e, gradient %.1e (relative to the largest entry)" % (rel(1), rel(2))
        print("LDS budget %-8s " % kb + "  ".join("B=%s %.4f ms" % (B, v[0]) for B, v in d.items()) + dev)
