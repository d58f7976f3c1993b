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


def oracle_gd_replay(name, iters, predict=False, info=None):
    """(loss_trace [iters], parameter_trace [iters, p]) of the oracle's loop started at the stored P[0]; entry k is what the reference stores as L[k+1], P[k+1].
    predict: every solve after the first starts from the GUARDED first-order prediction of its solution (ipopt_ms.guarded_start: what IRLLoop / PDP_MS_PREDICT_GUARD do on
    the GPU) instead of the previous solution; info (a dict) receives the rows at which the guard rejected a prediction and the Newton iterations spent."""
    from oracle import ipopt_ms, models, pdp_oracle as po
    d = np.load(os.path.join(GOLDEN, "demos_%s.npz" % name))
    h = np.load(os.path.join(GOLDEN, "irltrace_head_%s.npz" % name))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    n_demo, T = d["control"].shape[0], d["control"].shape[1]
    theta, lr = h["param"][0].copy(), float(h["lr"])
    warm = [None] * n_demo
    losses, params = [], []
    theta_prev, rejected, newton = None, [], 0
    for k in range(iters):
        loss, dp = 0.0, np.zeros(oc.p)
        for i in range(n_demo):
            start = warm[i]
            if predict and start is not None:
                start, rej = ipopt_ms.guarded_start(oc, d["state"][i, 0], *warm[i], theta_prev, theta - theta_prev)
                if rej:
                    rejected.append((k, i))
            s = ipopt_ms.solve(oc, d["state"][i, 0], T, theta, tol=1e-10, warm=start)
            newton += s["iterations"]
            warm[i] = (s["state_traj_opt"], s["control_traj_opt"], s["costate_traj_opt"])
            aux = oc.getAuxSys(s["state_traj_opt"], s["control_traj_opt"], s["costate_traj_opt"], theta)
            sol = po.lqr_from_aux(aux, oc.n, oc.p, T)
            l, g = po.irl_loss_grad(s["state_traj_opt"], s["control_traj_opt"], d["state"][i], d["control"][i], sol["state_traj_opt"], sol["control_traj_opt"])
            loss, dp = loss + l, dp + g
        theta_prev = theta
        theta = theta - lr * dp / n_demo
        losses.append(loss / n_demo)
        params.append(theta.copy())
    if info is not None:
        info.update(rejected=rejected, newton_iterations=newton)
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


@pytest.mark.parametrize("name,iters,rejections", [("rocket", 40, [(1, 0), (2, 0)]), ("cartpole", 30, [(4, 2)])])
def test_oracle_loop_with_guarded_prediction_replays_the_stored_trace(name, iters, rejections):
    """the loop as the device runs it - every solve from the guarded first-order prediction - stays on the stored trace, and the guard fires exactly where the
    prediction is worse than no prediction: rocket rows 1 and 2 (a 1 % parameter step across sensitivities of order 1e2), cart-pole row 4 demo 2"""
    h = np.load(os.path.join(GOLDEN, "irltrace_head_%s.npz" % name))
    info = {}
    L, P = oracle_gd_replay(name, iters, predict=True, info=info)
    assert info["rejected"] == rejections, info
    assert (np.abs(L - h["loss"][1:iters + 1]) / np.abs(h["loss"][1:iters + 1])).max() <= 2e-8
    assert np.abs(P - h["param"][1:iters + 1]).max() <= 1e-9


def test_unguarded_prediction_leaves_the_stored_rocket_trace():
    """why the guard exists: the reference's own rocket run, row 0 -> 1.  The unguarded prediction has ~50 times the KKT error of the previous solution, and Newton's
    method started there ends in another stationary point: loss 10289.857 where IPOPT (from the all-zero guess) stored 1301.237 - which the plain warm start reproduces."""
    from oracle import ipopt_ms, models, pdp_oracle as po
    d = np.load(os.path.join(GOLDEN, "demos_rocket.npz"))
    h = np.load(os.path.join(GOLDEN, "irltrace_head_rocket.npz"))
    st = models.IRL_SETUP["rocket"]
    oc = po.make_oc(models.REGISTRY["rocket"](**st["kwargs"]), st["dt"])
    T, x0 = d["control"].shape[1], d["state"][0, 0]
    th0, th1 = h["param"][0], h["param"][1]
    s0 = ipopt_ms.solve(oc, x0, T, th0, tol=1e-10)
    sol0 = (s0["state_traj_opt"], s0["control_traj_opt"], s0["costate_traj_opt"])
    pred = ipopt_ms.predict_start(oc, *sol0, th0, th1 - th0)
    e_pred, e_plain = ipopt_ms.scaled_kkt_error(oc, *pred, th1), ipopt_ms.scaled_kkt_error(oc, *sol0, th1)
    assert e_pred > 10 * e_plain
    start, rejected = ipopt_ms.guarded_start(oc, x0, *sol0, th0, th1 - th0)
    assert rejected

    def loss_of(s):
        return float(np.linalg.norm(s["state_traj_opt"] - d["state"][0]) ** 2 + np.linalg.norm(s["control_traj_opt"] - d["control"][0]) ** 2)
    assert abs(loss_of(ipopt_ms.solve(oc, x0, T, th1, tol=1e-10, warm=start)) - h["loss"][2]) <= 1e-9 * h["loss"][2]
    wrong = loss_of(ipopt_ms.solve(oc, x0, T, th1, tol=1e-10, warm=pred))
    assert abs(wrong - 10289.857357) <= 1e-3, wrong
