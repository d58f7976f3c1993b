#!/usr/bin/env python3
"""Headline kernel (oc_pdp_fused3_kernel, quadrotor, T = 50, B = 1024): HIP-event time of variant builds against the shipped library, and whether their outputs are the
shipped ones bit for bit.  Usage: python probes/f3_variants_timing.py label=path/to/lib.so ...   (libraries built beforehand, e.g. into probes/_build/)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import bench                      # noqa: E402
from pdp_amd import runtime, zoo      # noqa: E402

B = 1024
x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
libs = [("shipped", zoo.get("quadrotor", "irl"))] + [(a.split("=", 1)[0], runtime.ModelLib(a.split("=", 1)[1])) for a in sys.argv[1:]]
ref = None
rows = {k: [] for k, _ in libs}
outs = {}
for rnd in range(3):                       # interleaved rounds: clock / thermal drift hits every variant alike
    for k, mdl in libs:
        bufs = {}
        fn = lambda: mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs)
        rows[k].append(bench._event_ms(torch, fn, reps=20, warm=3))
        outs[k] = {a: b.clone() for a, b in fn().items() if a in ("loss", "grad", "x", "lam")}
for k, _ in libs:
    same = all(bool((outs[k][a] == outs["shipped"][a]).all()) for a in outs[k])
    print("%-28s %s ms   (min %.4f)   outputs bit-identical to the shipped build: %s" % (k, "  ".join("%.4f" % v for v in rows[k]), min(rows[k]), same))
