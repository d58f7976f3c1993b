"""GPU: the device-resident loops replay the reference's STORED gradient-descent traces (round-4 verdict, item 1).

The only multi-iteration truth the reference ships are loss_trace / parameter_trace of its IRL runs (Examples/IRL/<sys>/data/PDP_results_trial_0.mat: real CasADi + IPOPT;
loop: Examples/IRL/cartpole/cartpole_PDP.py:45-82, Examples/IRL/quadrotor/uav_PDP.py:52-62 - ocSolver from the all-zero guess on every demonstration, getAuxSys, lqrSolver,
chain rule, batch mean, out-of-place update).  tests/golden/irltrace_head_<sys>.npz holds the first 202 consecutive rows (make_fixtures.py).  IRLLoop (pdp_amd/irl.py) is
started at the stored P[0] with the stored learning rate - cold multiple-shooting solve, then every solve from the predicted start, gradient unit, one-launch update, eager
and as a hipGraph - and must write the stored rows: entry k of its traces is what the reference stored as L[k+1], P[k+1].

Stated tolerances (margins recorded): rows 1..3: loss 1e-9 relative, parameter 1e-7 (the gradient tolerance against IPOPT's traces) x lr x the
largest gradient entry x rows; all 200 rows: loss 2e-8 relative or 5e-9 absolute (IPOPT's termination noise in the stored rows; the oracle's own replay, tests/test_oracle_gd_replay.py,
is at 3e-9 over its 100), parameter 1e-9 absolute at every row, i.e. the loop lands on the stored P[200]."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS = 200          # (the fixtures hold rows 0 .. 201)


def _check(margins, label, L, P, h, lr):
    n = L.shape[0]
    Ls, Ps = h["loss"][1:n + 1], h["param"][1:n + 1]
    rel = np.abs(L - Ls) / np.abs(Ls)
    err = np.abs(P - Ps).max(axis=1)
    g = np.abs(h["param"][:n] - h["param"][1:n + 1]).max(axis=1) / lr                  # largest entry of the stored gradient of each row
    margins.check("%s: loss_trace rows 1..3 (relative)" % label, rel[:3].max(), 1e-9)
    margins.check("%s: parameter_trace rows 1..3 (in units of 1e-7 x lr x largest gradient entry x rows)" % label,
                  (err[:3] / (1e-7 * lr * np.maximum.accumulate(g[:3]) * np.arange(1, 4))).max(), 1.0)
    # every row: 2e-8 relative, or 5e-9 absolute where the loss itself has become small (quadrotor rows 170 .. 200: loss 0.04 - the stored values carry the error of
    # IPOPT's own termination, tol = 1e-8 on the states, which enters the loss as 2 |x - x_demo| 1e-8 whatever the loss)
    margins.check("%s: loss_trace all %d rows (in units of max(2e-8 relative, 5e-9 absolute))" % (label, n), (np.abs(L - Ls) / np.maximum(2e-8 * np.abs(Ls), 5e-9)).max(), 1.0)
    margins.check("%s: parameter_trace all %d rows (absolute)" % (label, n), err.max(), 1e-9)
    margins.check("%s: lands on the stored P[%d] (absolute)" % (label, n), err[-1], 1e-9)


@pytest.mark.parametrize("name", ["cartpole", "quadrotor", "rocket", "pendulum", "robotarm"])
@pytest.mark.parametrize("mode", ["graph", "eager"])
def test_irl_loop_replays_the_stored_trace(golden_dir, margins, name, mode):
    from pdp_amd import zoo
    from pdp_amd.irl import IRLLoop
    d = np.load(os.path.join(golden_dir, "demos_%s.npz" % name))
    h = np.load(os.path.join(golden_dir, "irltrace_head_%s.npz" % name))
    lr = float(h["lr"])
    mdl = zoo.get(name, "irl")
    loop = IRLLoop(mdl, d["state"], d["control"], h["param"][0], lr, record="full", max_steps=ITERS)
    assert loop.run(ITERS, graphed=(mode == "graph")) == ITERS
    r = loop.results()
    assert r["iterations"] == ITERS and r["unconverged_solves"] == 0 and r["riccati_trouble"] == 0, r
    _check(margins, "IRLLoop (%s) replay of the stored %s trace from P[0]" % (mode, name), r["loss_trace"], r["parameter_trace"], h, lr)


def test_unguarded_prediction_leaves_the_stored_rocket_trace():
    """why IRLLoop guards its predicted starts (PDP_MS_PREDICT_GUARD): without the guard the loop leaves the reference's rocket trace at its SECOND iteration - the
    unguarded prediction across the first parameter step ends in another stationary point, loss 10289.857 where IPOPT stored 1301.237; the oracle's unguarded loop does
    exactly the same (tests/test_oracle_gd_replay.py), i.e. this is the algorithm's behaviour, not a kernel defect"""
    from pdp_amd import zoo
    from pdp_amd.irl import IRLLoop
    d = np.load(os.path.join(ROOT, "tests", "golden", "demos_rocket.npz"))
    h = np.load(os.path.join(ROOT, "tests", "golden", "irltrace_head_rocket.npz"))
    loop = IRLLoop(zoo.get("rocket", "irl"), d["state"], d["control"], h["param"][0], float(h["lr"]), max_steps=2, guard=False)
    loop.run(2, graphed=False)
    L = loop.results()["loss_trace"]
    assert abs(L[0] - h["loss"][1]) <= 1e-9 * h["loss"][1] and abs(L[1] - 10289.857357) <= 1e-3, L
    guarded = IRLLoop(zoo.get("rocket", "irl"), d["state"], d["control"], h["param"][0], float(h["lr"]), max_steps=2)
    guarded.run(2, graphed=False)
    L = guarded.results()["loss_trace"]
    assert abs(L[1] - h["loss"][2]) <= 1e-9 * h["loss"][2], L


def test_short_graphed_runs_do_not_overshoot():
    """run(n, graphed=True) with n below the capture warm-up count does n iterations (round-4 advice: the warm-up used to run unconditionally)"""
    from pdp_amd import zoo
    from pdp_amd.irl import IRLLoop
    d = np.load(os.path.join(ROOT, "tests", "golden", "demos_cartpole.npz"))
    h = np.load(os.path.join(ROOT, "tests", "golden", "irltrace_head_cartpole.npz"))
    for n in (1, 2, 3, 4):
        loop = IRLLoop(zoo.get("cartpole", "irl"), d["state"], d["control"], h["param"][0], float(h["lr"]), max_steps=n)
        assert loop.run(n, graphed=True) == n
        r = loop.results()
        assert r["iterations"] == n and np.abs(r["parameter_trace"][n - 1] - h["param"][n]).max() <= 1e-11


@pytest.mark.parametrize("name,extra", [("cartpole", ["--graph"]), ("quadrotor", ["--graph", "--record", "primal"]), ("rocket", [])])
def test_irl_example_replays_the_stored_trace(golden_dir, margins, tmp_path, name, extra):
    """examples/irl_pdp.py (the counterpart of the reference's driver: OCSys built through the class surface, results saved in the reference's .mat schema) started at the
    stored P[0]: device loop as a hipGraph (--graph) and the host-driven loop (ocsolver.solve_batch + pdp_grad_batch, mean on the host)"""
    import scipy.io as sio
    h = np.load(os.path.join(golden_dir, "irltrace_head_%s.npz" % name))
    lr = float(h["lr"])
    out = str(tmp_path / "r.mat")
    cmd = [sys.executable, os.path.join(ROOT, "examples", "irl_pdp.py"), "--system", name, "--iters", str(ITERS), "--lr", repr(lr), "--init",
           ",".join(repr(float(v)) for v in h["param"][0]), "--out", out] + extra
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "warning" not in r.stdout, r.stdout[-3000:]
    res = sio.loadmat(out)["results"][0, 0]
    L, P = res["loss_trace"].flatten(), np.asarray(res["parameter_trace"], dtype=float).reshape(ITERS, -1)
    _check(margins, "examples/irl_pdp.py %s %s replay of the stored trace from P[0]" % (name, " ".join(extra) or "(host-driven loop)"), L, P, h, lr)
