#!/usr/bin/env python3
"""Fixture for OCSys.ocSolver WITH finite bounds (reference PDP/PDP.py:141-168 passes lbw / ubw to IPOPT; no stored reference run uses them, and
IPOPT is not available here): the pendulum swing-up of the IRL example (T = 20, dt = 0.1, theta = the demos' true parameter) with
    |u_t| <= 12  (the unconstrained optimum uses up to 28.5)   and   dq_t <= 6  (it reaches 11)
solved by an INDEPENDENT method - scipy's SLSQP on the single-shooting problem (controls only, state bounds as nonlinear inequalities), polished from
several starts, tolerances 1e-13 - on the oracle's sympy model (oracle/models.py).  Stored: x, u, cost, and which bounds are active."""
import os
import sys
import numpy as np
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import models, pdp_oracle as po       # noqa: E402

d = np.load(os.path.join(HERE, "demos_pendulum.npz"))
st = models.IRL_SETUP["pendulum"]
oc = po.make_oc(models.REGISTRY["pendulum"](**st["kwargs"]), st["dt"])
theta, T, x0 = d["true_parameter"], 20, d["state"][0, 0]
UMAX, VMAX = 12.0, 6.0


def rollout(u):
    x = np.zeros((T + 1, 2))
    x[0] = x0
    c = 0.0
    for t in range(T):
        c += float(oc.path_cost_fn(x[t], u[t:t + 1], theta))
        x[t + 1] = np.asarray(oc.dyn_fn(x[t], u[t:t + 1], theta), float).reshape(-1)
    return x, c + float(oc.final_cost_fn(x[T], theta))


obj = lambda u: rollout(u)[1]
cons = [{"type": "ineq", "fun": lambda u: VMAX - rollout(u)[0][1:, 1]}]
best = None
for seed in range(6):
    u0 = np.clip(d["control"][0, :, 0] * (0.3 + 0.1 * seed), -UMAX, UMAX) if seed else np.zeros(T)
    r = minimize(obj, u0, method="SLSQP", bounds=[(-UMAX, UMAX)] * T, constraints=cons, options={"ftol": 1e-15, "maxiter": 2000})
    for _ in range(3):
        r = minimize(obj, r.x, method="SLSQP", bounds=[(-UMAX, UMAX)] * T, constraints=cons, options={"ftol": 1e-16, "maxiter": 2000})
    x, c = rollout(r.x)
    print("start %d: cost %.12f  max|u| %.6f  max dq %.6f  %s" % (seed, c, np.abs(r.x).max(), x[:, 1].max(), r.message))
    if best is None or c < best[1] - 1e-9:
        best = (r.x.copy(), c, x)
u, c, x = best
np.savez_compressed(os.path.join(HERE, "bounded_oc_pendulum.npz"), x0=x0, theta=theta, T=T, dt=st["dt"], umax=UMAX, vmax=VMAX, control=u[:, None], state=x, cost=c,
                    active_u=np.abs(np.abs(u) - UMAX) < 1e-7, active_v=np.abs(x[1:, 1] - VMAX) < 1e-7)
print("stored: cost %.12f, active control bounds %d, active state bounds %d (unconstrained optimum: cost %.6f)" %
      (c, int((np.abs(np.abs(u) - UMAX) < 1e-7).sum()), int((np.abs(x[1:, 1] - VMAX) < 1e-7).sum()), float(d["cost"][0])))


# ---- round 4: a control-bounded cart-pole (Examples/IRL/cartpole setting, first stored demo, T = 30): |u_t| <= 10 where the unconstrained optimum uses up to 18.3.
# Same independent method (SLSQP on the single-shooting problem, several starts, polished); the x0 of this case has its FIRST state component exactly on a
# state bound that is never active afterwards (x >= x0[0] - 0: the cart may not move left of where it starts is NOT imposed; instead the bound q <= q0 + 10 is
# inactive) - the test uses it for "an initial state ON a state bound is accepted" with the bound  dq >= dq0  (= 0: the pole starts at rest) relaxed as IPOPT relaxes bounds.
def cartpole_case():
    dc = np.load(os.path.join(HERE, "demos_cartpole.npz"))
    stc = models.IRL_SETUP["cartpole"]
    occ = po.make_oc(models.REGISTRY["cartpole"](**stc["kwargs"]), stc["dt"])
    th, Tc, x0c = dc["true_parameter"], dc["control"].shape[1], dc["state"][0, 0]
    UM = 10.0

    def roll(u):
        x = np.zeros((Tc + 1, 4))
        x[0] = x0c
        c = 0.0
        for t in range(Tc):
            c += float(occ.path_cost_fn(x[t], u[t:t + 1], th))
            x[t + 1] = np.asarray(occ.dyn_fn(x[t], u[t:t + 1], th), float).reshape(-1)
        return x, c + float(occ.final_cost_fn(x[Tc], th))
    best = None
    for seed in range(6):
        u0 = np.clip(dc["control"][0, :, 0] * (1.0 - 0.15 * seed), -UM, UM) if seed else np.zeros(Tc)
        r = minimize(lambda u: roll(u)[1], u0, method="SLSQP", bounds=[(-UM, UM)] * Tc, options={"ftol": 1e-15, "maxiter": 3000})
        for _ in range(3):
            r = minimize(lambda u: roll(u)[1], r.x, method="SLSQP", bounds=[(-UM, UM)] * Tc, options={"ftol": 1e-16, "maxiter": 3000})
        x, c = roll(r.x)
        print("cart-pole start %d: cost %.12f  max|u| %.6f  %s" % (seed, c, np.abs(r.x).max(), r.message))
        if best is None or c < best[1] - 1e-9:
            best = (r.x.copy(), c, x)
    u, c, x = best
    np.savez_compressed(os.path.join(HERE, "bounded_oc_cartpole.npz"), x0=x0c, theta=th, T=Tc, dt=stc["dt"], umax=UM, control=u[:, None], state=x, cost=c,
                        active_u=np.abs(np.abs(u) - UM) < 1e-7)
    print("stored cart-pole: cost %.12f, active control bounds %d (unconstrained optimum: cost %.6f)" % (c, int((np.abs(np.abs(u) - UM) < 1e-7).sum()), float(dc["cost"][0])))


if "--cartpole" in sys.argv or True:
    cartpole_case()
