#!/usr/bin/env python3
"""What the predicted start costs and buys at C3 (quadrotor, T = 50, B = 1024, per-sample theta = theta* (1 +- 2 %)): HIP-event times of
  (1) the solve from the previous solution (plain warm start),
  (2) the solve from a point predicted BEFOREHAND (pdp_oc_predict_batched output handed over as the warm point: the pure cost of the iterations that remain),
  (3) the solve with the prediction applied inside the launch (PDP_MS_PREDICT), with and without the multiplier part,
  (4) the gradient unit without / with its sensitivity outputs.
(3) - (2) is what reading the sensitivities costs inside the solver."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                      # noqa: E402
import bench                      # noqa: E402
from pdp_amd import runtime as rt, zoo      # noqa: E402

rng = np.random.default_rng(0)
if "cartpole" in sys.argv[1:]:          # C2 (bench.py's setting): 256 cart-pole problems, theta* +- 0.05
    mdl = zoo.get("cartpole", "irl")
    B, T = 256, 50
    th = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
    x0 = np.zeros((B, 4))
    x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    x0 = rt.dev(x0)
    th1 = rt.dev(th[None] + rng.uniform(-0.05, 0.05, (B, 7)))
else:
    mdl = zoo.get("quadrotor", "irl")
    B, T = 1024, 50
    th = np.array(bench.THETA)
    x0 = rt.dev(bench.synth_inputs(B, 5)[0])
    th1 = rt.dev(th[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, th.size))))
print("%s, B = %d, T = %d" % (mdl.name if hasattr(mdl, "name") else "model", B, T))
demo = mdl.oc_solve_ms(x0, th, T)
warm = (demo["state"], demo["control"], demo["costate"])
sens = mdl.oc_pdp_grad(demo["control"], th, demo["state"], demo["control"], x=demo["state"], lam=demo["costate"], want_sens=True, want_riccati=True)
dth = (th1 - torch.as_tensor(th, device="cuda").view(1, -1)).contiguous()
pred_pt = mdl.oc_predict(*warm, dth, sens["dxdp"], sens["dudp"], sens["riccati"])
pred_xu = mdl.oc_predict(*warm, dth, sens["dxdp"], sens["dudp"])


def timed(label, make):
    copies = [make() for _ in range(8)]
    fn = lambda: mdl.oc_solve_ms(x0, th1, T, consume_warm=True, **copies.pop())
    ms = bench._event_ms(torch, fn, reps=5, warm=1)
    r = mdl.oc_solve_ms(x0, th1, T, consume_warm=True, **make())
    it = r["iterations"].double()
    print("%-78s %.4f ms   iterations mean %.2f max %d   converged %d" % (label, ms, float(it.mean()), int(it.max()), int(r["converged"].sum())))


cl = lambda tup: tuple(a.clone() for a in tup)
timed("(1) plain warm start (previous solution)", lambda: dict(warm=cl(warm)))
timed("(2) from the point predicted beforehand (x, u, lambda)", lambda: dict(warm=cl(pred_pt)))
timed("(2') from the point predicted beforehand (x, u only)", lambda: dict(warm=cl(pred_xu)))
timed("(3) prediction inside the launch, x, u, lambda (181 MB of sensitivities read)", lambda: dict(warm=cl(warm), predict=dict(dtheta=dth, dxdp=sens["dxdp"], dudp=sens["dudp"], riccati=sens["riccati"])))
timed("(3') prediction inside the launch, x, u only (64 MB)", lambda: dict(warm=cl(warm), predict=dict(dtheta=dth, dxdp=sens["dxdp"], dudp=sens["dudp"])))
rec = mdl.oc_pdp_grad(demo["control"], th, demo["state"], demo["control"], x=demo["state"], lam=demo["costate"], want_predict_record=True)["predict_record"]
timed("(3'') prediction inside the launch from the packed fp32 record (%d MB)" % (rec.numel() * 4 // 2 ** 20), lambda: dict(warm=cl(warm), predict=dict(dtheta=dth, record=rec)))
recp = mdl.oc_pdp_grad(demo["control"], th, demo["state"], demo["control"], x=demo["state"], lam=demo["costate"], want_predict_record="primal")["predict_record"]
timed("(3p) ... from the X | U part of the record only (PDP_MS_PREDICT_PRIMAL, %d MB touched)" % (B * T * (mdl.n + mdl.m) * mdl.p * 4 // 2 ** 20),
      lambda: dict(warm=cl(warm), predict=dict(dtheta=dth, record=recp, primal=True)))
sol = mdl.oc_solve_ms(x0, th1, T, warm=cl(pred_pt), consume_warm=True)
b0, b1, b2, b3, b4 = {}, {}, {}, {}, {}
g = lambda **kw: mdl.oc_pdp_grad(sol["control"], th1, demo["state"], demo["control"], x=sol["state"], lam=sol["costate"], **kw)
print("(4) gradient unit: plain %.4f ms | + dxdp, dudp %.4f ms | + dxdp, dudp, riccati %.4f ms | + the fp32 record only %.4f ms | + its X | U part only %.4f ms" % (
    bench._event_ms(torch, lambda: g(buffers=b0)), bench._event_ms(torch, lambda: g(buffers=b1, want_sens=True)), bench._event_ms(torch, lambda: g(buffers=b2, want_sens=True, want_riccati=True)),
    bench._event_ms(torch, lambda: g(buffers=b3, want_predict_record=True)), bench._event_ms(torch, lambda: g(buffers=b4, want_predict_record="primal"))))
