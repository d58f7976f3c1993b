#!/usr/bin/env python3
"""Rollout pre-pass of SysID.step (round 4): batches with several trajectories per SIMD roll out beforehand with ONE LANE per trajectory (sysid_integrate_kernel into a
workspace) and run the fused kernel on the given trajectories, instead of one uniform rollout per wavefront inside it.  PDP_SYSID_PREPASS = 0 / 1 forces the mode (read once
per process: one subprocess per setting); "default" is the library's rule.  Prints the time of SysID.step per batch size and the deviation of loss / gradient between the modes."""
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
for B in (1024, 2048, 4096, 8192, 16384):
    T = 100
    u = rng.uniform(-1, 1, (B, T, 4)) + 2.5
    x0 = np.tile(np.array([-8, -6, 9.0, 0, 0, 0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0, 0, 0]), (B, 1))
    x0[:, :3] += rng.standard_normal((B, 3))
    xobs = mdl.sysid_integrate(x0, u, np.array([1, 1, 1, 1, .4]))
    th = np.array([1.1, 0.95, 1.08, 1.03, 0.38])
    ud = torch.as_tensor(u, device="cuda")
    ms = bench._event_ms(torch, lambda: mdl.sysid_step(ud, xobs, th), reps=7, warm=2)
    l, g = mdl.sysid_step(ud, xobs, th)
    out[str(B)] = [ms, l.cpu().numpy().tolist(), g.cpu().numpy().tolist()]
print("RESULT " + json.dumps(out))
""" % ROOT

if __name__ == "__main__":
    import json
    import numpy as np
    res = {}
    for mode in ("0", "1", "default"):
        env = dict(os.environ)
        env.pop("PDP_SYSID_PREPASS", None)
        if mode != "default":
            env["PDP_SYSID_PREPASS"] = mode
        r = subprocess.run([sys.executable, "-c", WORKER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(mode, "FAILED", r.stdout[-800:])
            continue
        res[mode] = json.loads(line[0][7:])
    ref = res.get("0")
    for mode, d in res.items():
        dev = ""
        if ref:
            dl = max(np.abs(np.array(d[B][1]) - np.array(ref[B][1])).max() / np.abs(np.array(ref[B][1])).max() for B in d)
            dg = max(np.abs(np.array(d[B][2]) - np.array(ref[B][2])).max() / np.abs(np.array(ref[B][2])).max() for B in d)
            dev = "   largest deviation from the in-kernel rollout: loss %.1e, gradient %.1e (relative to the largest entry)" % (dl, dg)
        print("pre-pass %-8s " % mode + "  ".join("B=%s %.4f ms" % (B, v[0]) for B, v in d.items()) + dev)
