#!/usr/bin/env python3
"""Occupancy mode of sysid_step_kernel (round 4): C5a's total batch on ONE GPU (8192 quadrotor trajectories, T = 100) with the pool cut so that two
wavefronts share a SIMD (PDP_SYSID_ROWS overrides the rows per Jacobian pass; default: sysid_rows in csrc/pdp_model_kernels.h).  One subprocess per setting
(the override is read once per process).  Prints kernel time per setting and the deviation of loss / gradient from the full-pool run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from pdp_amd import JinEnv, zoo
mdl = zoo.get("quadrotor", "sysid")
rng = np.random.default_rng(0)
out = {}
for B in (1024, 2048, 4096, 8192):
    T = 100
    u = rng.uniform(-1, 1, (B, T, 4)) + 2.5
    x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
    xobs = mdl.sysid_integrate(x0, u, np.array([1, 1, 1, 1, .4]))
    th = np.array([1.1, 0.95, 1.08, 1.03, 0.38])
    ud = torch.as_tensor(u, device="cuda")
    ms = bench._event_ms(torch, lambda: mdl.sysid_step(ud, xobs, th), reps=7, warm=2)
    l, g = mdl.sysid_step(ud, xobs, th)
    out[str(B)] = [ms, float(l.double().sum()), g.double().abs().sum().item(), g[B // 2].cpu().numpy().tolist()]
print("RESULT " + json.dumps(out))
""" % ROOT

if __name__ == "__main__":
    import json
    res = {}
    for rows in ("32", "16", "10", "8", "0"):
        r = subprocess.run([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, PDP_SYSID_ROWS=rows))
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(rows, "FAILED", r.stdout[-800:])
            continue
        res[rows] = json.loads(line[0][7:])
    ref = res.get("32")
    for rows, d in res.items():
        print("rows %-3s (0 = default rule): " % rows + "  ".join("B=%s %.4f ms" % (B, v[0]) for B, v in d.items()) +
              ("   max rel dev of sum(loss) from rows=32: %.1e, gradient row: %.1e" % (max(abs(d[B][1] - ref[B][1]) / abs(ref[B][1]) for B in d),
                                                                                    max(max(abs(a - b) for a, b in zip(d[B][3], ref[B][3])) / max(abs(b) for b in ref[B][3]) for B in d)) if ref else ""))
