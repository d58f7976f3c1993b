"""CPU: the oracle replays the reference's stored gradient-descent traces.

The reference's IRL drivers (Examples/IRL/cartpole/cartpole_PDP.py:45-82, Examples/IRL/quadrotor/uav_PDP.py:52-62) run, per iteration, ocSolver (IPOPT from the all-zero
guess) on every demonstration, getAuxSys, lqrSolver, the chain rule, the batch mean and the out-of-place update `current_parameter = current_parameter - lr * dp`, and store
loss_trace / parameter_trace (real CasADi + IPOPT on the author's machine).  tests/golden/irltrace_head_<sys>.npz holds the first 202 consecutive rows of trial 0.  Here the
oracle (oracle/ipopt_ms.py: IPOPT's algorithm restated; oracle/pdp_oracle.py: getAuxSys / lqrSolver / chain rule in the reference's order) runs that LOOP from P[0] -
cold solve first, every later solve warm-started from the previous solution - and must reproduce the stored rows, not just single iterations at stored parameters.
Indexing (make_fixtures.py): loss(P[k]) == L[k+1], P[k+1] == P[k] - lr * grad(P[k]).  The stored rows carry IPOPT's own termination noise (tolerance 1e-8):
achieved over 200 iterations - loss 3e-9 relative, parameter 7e-11 absolute."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def oracle_gd_replay(name, iters):
    """(loss_trace [iters], parameter_trace [iters, p]) of the oracle's loop started at the stored P[0]; entry k is what the reference stores as L[k+1], P[k+1]"""
    from oracle import ipopt_ms, models, pdp_oracle as po
    d = np.load(os.path.join(GOLDEN, "demos_%s.npz" % name))
    h = np.load(os.path.join(GOLDEN, "irltrace_head_%s.npz" % name))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    n_demo, T = d["control"].shape[0], d["control"].shape[1]
    theta, lr = h["param"][0].copy(), float(h["lr"])
    warm = [None] * n_demo
    losses, params = [], []
    for k in range(iters):
        loss, dp = 0.0, np.zeros(oc.p)
        for i in range(n_demo):
            s = ipopt_ms.solve(oc, d["state"][i, 0], T, theta, tol=1e-10, warm=warm[i])
            warm[i] = (s["state_traj_opt"], s["control_traj_opt"], s["costate_traj_opt"])
            aux = oc.getAuxSys(s["state_traj_opt"], s["control_traj_opt"], s["costate_traj_opt"], theta)
            sol = po.lqr_from_aux(aux, oc.n, oc.p, T)
            l, g = po.irl_loss_grad(s["state_traj_opt"], s["control_traj_opt"], d["state"][i], d["control"][i], sol["state_traj_opt"], sol["control_traj_opt"])
            loss, dp = loss + l, dp + g
        theta = theta - lr * dp / n_demo
        losses.append(loss / n_demo)
        params.append(theta.copy())
    return np.array(losses), np.array(params)


@pytest.mark.parametrize("name,iters", [("cartpole", 100), ("quadrotor", 100), ("rocket", 100), ("pendulum", 40), ("robotarm", 40)])
def test_oracle_loop_replays_the_stored_trace(name, iters):
    h = np.load(os.path.join(GOLDEN, "irltrace_head_%s.npz" % name))
    L, P = oracle_gd_replay(name, iters)
    Ls, Ps = h["loss"][1:iters + 1], h["param"][1:iters + 1]
    rel = np.abs(L - Ls) / np.abs(Ls)
    err = np.abs(P - Ps).max(axis=1)
    assert rel[:3].max() <= 1e-9, (name, rel[:3])                         # rows 1..3
    assert err[:3].max() <= 1e-11, (name, err[:3])
    assert rel.max() <= 2e-8, (name, rel.max())                           # stored loss_trace to <= 2e-8
    assert err[-1] <= 1e-9 and err.max() <= 1e-9, (name, err.max())      # lands on the stored P[iters]
