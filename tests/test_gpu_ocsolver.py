"""GPU: the batched OC solver that stands where the reference calls IPOPT (OCSys.ocSolver, PDP.py:121-220).
Known answers: the optima IPOPT found on the author's machine (tests/golden/demos_*.npz) and the stored IRL traces -
here the WHOLE iteration (OC solve -> aux system -> Riccati -> gradient) runs on the GPU; the oracle (oracle/ipopt_ms.py) appears
only as the checker of the multiple-shooting kernel's iteration log."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def load(golden_dir, f):
    return np.load(os.path.join(golden_dir, f))


def make_oc(name):
    from pdp_amd import PDP, zoo
    from pdp_amd.sx import vertcat
    env, dt = zoo.make_env(name, "irl")
    oc = PDP.OCSys(name)
    oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar))
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U)
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    return oc


@pytest.mark.parametrize("name", ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
def test_oc_solver_reproduces_stored_ipopt_optimum(golden_dir, name):
    """Cold start on all five systems, called like the reference calls ocSolver (no starting controls): the multiple-shooting NLP from
    the all-zero guess (PDP.py:155,166) iterated the way IPOPT does lands in the optimum IPOPT stored - including the non-convex rocket
    landing problem, where single shooting from u = 0 ends in another basin.  Every sample stays on the multiple-shooting path: robot arm demo 3,
    whose line search falls below alpha_min, goes through the kernel's feasibility restoration (round 3; before: single-shooting fallback)."""
    from pdp_amd import ocsolver
    d = load(golden_dir, "demos_%s.npz" % name)
    oc = make_oc(name)
    sol = ocsolver.solve_batch(oc, d["state"][:, 0], d["control"].shape[1], d["true_parameter"])
    assert bool(sol["converged"].all())
    assert bool(sol["method_ms"].all())
    restored = (sol["status"].cpu().numpy() & 128) != 0                      # PDP_MS_RESTORED
    assert (restored == ((np.arange(d["state"].shape[0]) == 3) if name == "robotarm" else np.zeros(d["state"].shape[0], bool))).all()
    x, u, lam, cost = (sol[k].cpu().numpy() for k in ("state", "control", "costate", "cost"))
    assert np.abs(cost - d["cost"]).max() <= 1e-9 * np.abs(d["cost"]).max()
    assert np.abs(x - d["state"]).max() <= 1e-6 * max(1, np.abs(d["state"]).max())
    assert np.abs(u - d["control"]).max() <= 1e-6 * max(1, np.abs(d["control"]).max())
    assert np.abs(lam - d["costate"]).max() <= 1e-6 * max(1, np.abs(d["costate"]).max())      # costate[t] = lambda_{t+1}, IPOPT lam_g sign


# (loss, gradient): about ten times what each system achieves (profiles/r03_parity_margins.txt).  Quadrotor, rocket and pendulum sit far inside BASELINE.md
# section 3's 1e-9 / 1e-7; the cart-pole and the robot arm are limited by the stored side of the comparison - IPOPT stopped those optima at its own 1e-8,
# and the swing-up / arm problems amplify that into 1e-8 (loss) / 3e-8 (gradient), the same figures the CPU oracle reaches on these rows
END2END_BOUNDS = {"cartpole": (1e-7, 5e-7), "quadrotor": (1e-12, 1e-9), "pendulum": (1e-11, 1e-8), "robotarm": (2e-8, 2e-7), "rocket": (1e-11, 1e-8)}


@pytest.mark.parametrize("name,rows", [("cartpole", [0, 3, 7]), ("quadrotor", [1, 5, 9]), ("pendulum", [0, 9]), ("robotarm", [0, 4]), ("rocket", [0, 3, 8])])
def test_end_to_end_irl_iteration_on_gpu_matches_stored_trace(golden_dir, margins, name, rows):
    """Examples/IRL/<sys>/<sys>_PDP.py loop body at the reference's own iterates theta_k: ocSolver -> getAuxSys -> lqrSolver ->
    chain rule, entirely on the GPU; reproduces loss_trace[k+1] and (p_k - p_{k+1})/lr stored by the reference."""
    from pdp_amd import ocsolver
    d = load(golden_dir, "demos_%s.npz" % name)
    tr = load(golden_dir, "irltrace_%s.npz" % name)
    oc = make_oc(name)
    T = d["control"].shape[1]
    for j in rows:
        th = tr["param"][j]
        sol = ocsolver.solve_batch(oc, d["state"][:, 0], T, th)       # cold, like the reference's loop (ocSolver from the zero guess at every iterate)
        assert bool(sol["converged"].all())
        out = oc.pdp_grad_batch(sol["control"], th, d["state"], d["control"], state_traj=sol["state"], costate_traj=sol["costate"])
        assert int(out["status"].sum()) == 0
        loss = float(out["loss"].mean())
        dp = out["grad"].mean(dim=0).cpu().numpy()
        gref = (tr["param"][j] - tr["param_next"][j]) / float(tr["lr"])
        # the stored trace comes from IPOPT optima (stop at its own 1e-8); this solve stops at 1e-10: the two ends of the comparison differ by what
        # IPOPT left - the bounds are END2END_BOUNDS (ten times the error each system achieved, profiles/r03_parity_margins.txt), never looser than
        # BASELINE.md section 3's 1e-9 / 1e-7 allow for an exact optimum
        bl, bg = END2END_BOUNDS[name]
        margins.check("GPU end to end (cold OC solve + gradient unit) vs stored IRL trace, %s row %d: loss (relative)" % (name, j),
                      abs(loss - tr["loss_next"][j]) / abs(tr["loss_next"][j]), bl)
        margins.check("GPU end to end (cold OC solve + gradient unit) vs stored IRL trace, %s row %d: gradient (relative to its largest entry)" % (name, j),
                      np.abs(dp - gref).max() / np.abs(gref).max(), bg)


def test_ocSolver_dropin_signature(golden_dir):
    """reference call `traj = oc.ocSolver(ini_state=..., horizon=..., auxvar_value=...)` and its dict keys (PDP.py:212-218)"""
    d = load(golden_dir, "demos_quadrotor.npz")
    oc = make_oc("quadrotor")
    traj = oc.ocSolver(ini_state=d["state"][0, 0], horizon=50, auxvar_value=d["true_parameter"])
    assert set(traj) == {"state_traj_opt", "control_traj_opt", "costate_traj_opt", "auxvar_value", "time", "horizon", "cost"}
    assert traj["state_traj_opt"].shape == (51, 13) and traj["control_traj_opt"].shape == (50, 4) and traj["costate_traj_opt"].shape == (50, 13)
    assert abs(traj["cost"].item() - d["cost"][0]) < 1e-9 * d["cost"][0]
    assert np.abs(traj["costate_traj_opt"] - d["costate"][0]).max() < 1e-6 * np.abs(d["costate"][0]).max()
    traj1 = oc.ocSolver(d["state"][0, 0], 50, d["true_parameter"], costate_option=1)
    assert np.abs(traj1["costate_traj_opt"] - traj["costate_traj_opt"]).max() < 1e-8 * np.abs(d["costate"][0]).max()


def test_oc_solve_entry_point_returns_a_kkt_point():
    """pdp_oc_solve_batched called directly (no batch-level globalisation on top): at convergence the outputs satisfy the conditions
    IPOPT solves - x is the rollout of u, lam the costate recursion along (x, u), |H_u| below the tolerance - and the reported
    cost / gradient norm / gains describe that point; the caller's initial controls are left untouched."""
    import torch
    oc = make_oc("pendulum")
    mdl = oc.model()
    rng = np.random.default_rng(5)
    B, T = 37, 20
    x0 = np.stack([rng.uniform(-0.5, 0.5, B), np.zeros(B)], axis=1)
    th = np.array([1.0, 1.0, 0.05, 10.0, 1.0]) * (1 + 0.1 * rng.uniform(-1, 1, (B, 5)))
    u0 = torch.zeros((B, T, 1), dtype=torch.float64, device="cuda")
    sol = mdl.oc_solve(x0, u0, th, tol=1e-9, max_iter=100, want_gains=True)
    assert bool(sol["converged"].all()) and 0 < sol["iterations"] < 100 and float(u0.abs().max()) == 0.0
    x, J = mdl.oc_rollout(x0, sol["control"], th)
    assert float((x - sol["state"]).abs().max()) <= 1e-12 and float((J - sol["cost"]).abs().max()) <= 1e-10 * float(J.abs().max())
    lam = mdl.oc_costate(sol["state"], sol["control"], th)
    assert float((lam - sol["costate"]).abs().max()) <= 1e-12 * max(1.0, float(lam.abs().max()))
    hu = mdl.oc_auxsys(sol["state"], sol["control"], sol["costate"], th, only=("dHu",))["dHu"]
    scale = 1 + sol["control"].abs().amax(dim=(1, 2))
    assert bool((hu.abs().amax(dim=(1, 2)) <= 1e-9 * scale).all())
    assert float((hu.abs().amax(dim=(1, 2)) - sol["grad_norm"]).abs().max()) <= 1e-12
    # the gains are the LQR feedback around the solution: rolling them out from a perturbed initial state is a descent start
    xp = x0 + 0.05 * rng.standard_normal(x0.shape)
    _, u_cl, J_cl = mdl.oc_rollout_feedback(xp, sol["control"], sol["state"], sol["gains"], torch.zeros((B,), dtype=torch.float64, device="cuda"), th)
    _, J_ol = mdl.oc_rollout(xp, sol["control"], th)
    assert bool((J_cl <= J_ol + 1e-12).all())


def _alpha_col(l):
    """the iteration log's alpha column: minus the test step length where a second-order-corrected step was taken (include/pdp_hip.h)"""
    return -l["alpha"] if l.get("soc_taken") else l["alpha"]


def _follows(golden_dir, name, demo, soc, rows=None):
    """rows: compare only the first `rows` iterations (a non-convex solve amplifies rounding differences: the cart-pole cold solves part from the restatement's digits
    after a dozen iterations - f agrees to 1e-10 at iteration 11, 1e-9 at 12, ... - and from its decisions later; both end in the stored optimum)"""
    from oracle import ipopt_ms, models, pdp_oracle as po
    from pdp_amd import zoo
    d = load(golden_dir, "demos_%s.npz" % name)
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    T = d["control"].shape[1]
    log = []
    ref = ipopt_ms.solve(oc, d["state"][demo, 0], T, d["true_parameter"], log=log, soc=soc)
    mdl = zoo.get(name, "irl")
    x0 = np.repeat(d["state"][demo:demo + 1, 0], 3, axis=0)                  # (a few copies: the last one is compared)
    sol = mdl.oc_solve_ms(x0, d["true_parameter"], T, tol=1e-10, log_rows=len(log) + 4, soc=soc)
    taken = sum(1 for l in log if l["soc_taken"])
    assert bool(sol["converged"].all()) and (sol["status"].cpu().numpy() == (1024 if taken else 0)).all()
    assert ref["iterations"] == len(log) and (rows is not None or (sol["iterations"].cpu().numpy() == ref["iterations"]).all())
    kl = sol["log"][2].cpu().numpy()
    for r, l in list(zip(kl, log))[:rows]:
        assert r[5] == _alpha_col(l), (name, l["it"], r[5], l["alpha"], l["soc_taken"])
        assert abs(r[4] - l["dw"]) <= 1e-12 * max(1.0, l["dw"])
        assert abs(r[1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"]))
        assert abs(r[7] - l["theta"]) <= 1e-9 * max(1.0, l["theta"]) and abs(r[2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
    sc = lambda a: max(1.0, np.abs(a).max())
    tol = 1e-9 if rows is None else 1e-7                                     # (two iterate paths into one optimum, each to the 1e-10 convergence test)
    assert np.abs(sol["state"][2].cpu().numpy() - ref["state_traj_opt"]).max() <= tol * sc(ref["state_traj_opt"])
    assert np.abs(sol["control"][2].cpu().numpy() - ref["control_traj_opt"]).max() <= tol * sc(ref["control_traj_opt"])
    assert np.abs(sol["costate"][2].cpu().numpy() - ref["costate_traj_opt"]).max() <= tol * sc(ref["costate_traj_opt"])
    assert abs(float(sol["cost"][2]) - d["cost"][demo]) <= 1e-9 * abs(d["cost"][demo])      # ... which is the one IPOPT stored
    return ref, log, d


@pytest.mark.parametrize("name", ["pendulum", "rocket", "quadrotor"])
def test_ms_kernel_follows_the_oracle_iteration_by_iteration(golden_dir, name):
    """pdp_oc_solve_ms_batched against oracle/ipopt_ms.py (the CPU restatement of IPOPT's algorithm on the reference's NLP) on the
    first stored demo: same number of iterations, same inertia corrections and step lengths at every iteration, objective /
    infeasibility columns of the iteration log equal to rounding, same solution."""
    _follows(golden_dir, name, 0, False)


@pytest.mark.parametrize("name,demo,tried,taken,iters,iters_plain,rows", [("cartpole", 0, 22, 7, 35, 45, 11), ("robotarm", 0, 1, 1, 3, 6, None), ("robotarm", 1, 4, 1, 5, 14, None),
                                                                          ("quadrotor", 0, 1, 0, 11, 11, None), ("rocket", 0, 2, 0, 10, 10, None)])
def test_ms_kernel_second_order_correction_follows_the_oracle(golden_dir, name, demo, tried, taken, iters, iters_plain, rows):
    """PDP_MS_WITH_SOC: the kernel's line search with IPOPT's second-order correction against the restatement's (solve(soc=True)) row by row - cart-pole and robot-arm
    demos TAKE corrected steps (minus alpha in the log, PDP_MS_SOC in the status), quadrotor and rocket try one or two, reject them and go on with the plain step halved,
    exactly as without the switch.  Cart-pole demo 0 (the small-system form of the kernel: the line search's trial pass split between the two waves): the first 11 rows -
    corrections taken at iterations 1, 3, 8, rejected at 5, 6, 9, 10 - then rounding takes the two non-convex solves apart (restatement 35 iterations, kernel 39; 45
    without the switch).  Every one of these solves ends in the optimum IPOPT stored."""
    ref, log, d = _follows(golden_dir, name, demo, True, rows=rows)
    assert ref["soc_steps"] == tried and sum(1 for l in log if l["soc_taken"]) == taken and ref["iterations"] == iters
    if rows is not None:
        assert sum(1 for l in log[:rows] if l["soc_taken"]) >= 3 and sum(1 for l in log[:rows] if l["soc"] and not l["soc_taken"]) >= 3
    assert abs(ref["cost"] - d["cost"][demo]) <= 1e-9 * abs(d["cost"][demo])
    from pdp_amd import zoo
    plain = zoo.get(name, "irl").oc_solve_ms(d["state"][demo:demo + 1, 0], d["true_parameter"], d["control"].shape[1], tol=1e-10)
    assert int(plain["iterations"][0]) == iters_plain and int(plain["status"][0]) == 0
    assert abs(float(plain["cost"][0]) - d["cost"][demo]) <= 1e-9 * abs(d["cost"][demo])


def test_ms_kernel_second_order_correction_after_a_restoration(golden_dir):
    """robot arm demo 3 with PDP_MS_WITH_SOC: 40 corrections tried around the restoration of iteration 8, two taken after it; 19 iterations instead of 21, the same stored optimum
    (to the 1e-10 convergence test: the last iterate sits 1.2e-9 from the stored controls where the plain iteration, one step later, sits 3e-13)."""
    from oracle import ipopt_ms, models, pdp_oracle as po
    from pdp_amd import zoo
    d = load(golden_dir, "demos_robotarm.npz")
    st = models.IRL_SETUP["robotarm"]
    oc = po.make_oc(models.REGISTRY["robotarm"](**st["kwargs"]), st["dt"])
    T = d["control"].shape[1]
    log = []
    ref = ipopt_ms.solve(oc, d["state"][3, 0], T, d["true_parameter"], tol=1e-10, log=log, soc=True)
    assert ref["iterations"] == 19 and ref["restorations"] == 1 and ref["soc_steps"] == 40
    out = zoo.get("robotarm", "irl").oc_solve_ms(d["state"][3:4, 0], d["true_parameter"], T, tol=1e-10, log_rows=32, soc=True)
    assert bool(out["converged"][0]) and int(out["status"][0]) == 128 + 1024 and int(out["iterations"][0]) == 19
    kl = out["log"][0].cpu().numpy()
    for r, l in enumerate(log):
        assert kl[r, 0] == l["it"] and kl[r, 4] == l["dw"] and kl[r, 5] == _alpha_col(l), (r, kl[r], l["dw"], l["alpha"])
        assert abs(kl[r, 1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(kl[r, 2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
    assert np.abs(out["control"][0].cpu().numpy() - d["control"][3]).max() <= 5e-9 and abs(float(out["cost"][0]) - d["cost"][3]) <= 1e-9 * d["cost"][3]


@pytest.mark.parametrize("name,demo,iters,status", [("robotarm", 1, 5, 1024), ("robotarm", 3, 19, 1152), ("quadrotor", 0, 11, 0)])
def test_second_order_correction_in_every_workgroup_shape(golden_dir, name, demo, iters, status):
    """the corrections' bookkeeping lives in the trajectory's LDS mailbox: one, two and four trajectories per workgroup (B = 3, 300, 1100 copies of a stored demo) give the same
    iterations, status and - bit for bit - the same solution (robot arm: corrections taken, one case around a restoration; quadrotor: tried and rejected, the plain step restored)"""
    from pdp_amd import zoo
    d = load(golden_dir, "demos_%s.npz" % name)
    mdl = zoo.get(name, "irl")
    T = d["control"].shape[1]
    ref = None
    for B in (3, 300, 1100):
        sol = mdl.oc_solve_ms(np.repeat(d["state"][demo:demo + 1, 0], B, axis=0), d["true_parameter"], T, tol=1e-10, soc=True)
        assert bool(sol["converged"].all()) and (sol["iterations"].cpu().numpy() == iters).all() and (sol["status"].cpu().numpy() == status).all()
        got = [sol[k].cpu().numpy() for k in ("state", "control", "costate")]
        for g in got:
            assert (g == g[:1]).all()                                        # every copy alike
        if ref is None:
            ref = got
        else:
            assert all(np.array_equal(a[0], b[0]) for a, b in zip(got, ref)), B


def test_ms_kernel_restoration_follows_the_oracle(golden_dir):
    """robot arm demo 3 (the one stored demo whose line search falls below alpha_min): the kernel's iteration log equals the restatement's row by row -
    eight Newton iterations, the restoration (step length 0 in the log), the least-squares multiplier reset, twelve more iterations - and ends in the optimum
    IPOPT stored; with PDP_MS_NO_RESTORATION the trajectory is returned with PDP_MS_RESTORATION where the restoration would have started."""
    from oracle import ipopt_ms, models, pdp_oracle as po
    from pdp_amd import zoo
    d = load(golden_dir, "demos_robotarm.npz")
    st = models.IRL_SETUP["robotarm"]
    oc = po.make_oc(models.REGISTRY["robotarm"](**st["kwargs"]), st["dt"])
    T = d["control"].shape[1]
    log = []
    ref = ipopt_ms.solve(oc, d["state"][3, 0], T, d["true_parameter"], tol=1e-10, log=log)
    mdl = zoo.get("robotarm", "irl")
    x0 = np.repeat(d["state"][3:4, 0], 5, axis=0)                            # a few copies: also through the TPW = 1 workgroup shape with company
    out = mdl.oc_solve_ms(x0, d["true_parameter"], T, tol=1e-10, log_rows=64)
    assert bool(out["converged"].all()) and (out["status"].cpu().numpy() == 128).all()
    assert (out["iterations"].cpu().numpy() == ref["iterations"]).all() and ref["restorations"] == 1
    kl = out["log"].cpu().numpy()[0]
    for r, l in enumerate(log):
        assert kl[r, 0] == l["it"] and kl[r, 4] == l["dw"] and kl[r, 5] == _alpha_col(l), (r, kl[r], l["dw"], l["alpha"])
        assert abs(kl[r, 1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(kl[r, 2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
        assert abs(kl[r, 3] - l["inf_du"]) <= 1e-8 * max(1.0, l["inf_du"])
    assert [l["alpha"] for l in log].index(0.0) == 8
    for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
        assert np.abs(out[k].cpu().numpy() - ref[kr][None]).max() <= 1e-9 * max(1.0, np.abs(ref[kr]).max())
    assert np.abs(out["state"].cpu().numpy()[0] - d["state"][3]).max() <= 1e-6 and abs(float(out["cost"][0]) - d["cost"][3]) <= 1e-9 * d["cost"][3]
    off = mdl.oc_solve_ms(x0, d["true_parameter"], T, tol=1e-10, log_rows=64, restoration=False)
    assert not bool(off["converged"].any()) and (off["status"].cpu().numpy() == 4).all() and (off["iterations"].cpu().numpy() == 8).all()


def test_ms_kernel_from_controls_follows_the_oracle_at_a_long_horizon(golden_dir):
    """PDP_MS_FROM_CONTROLS: the solve starts from a control guess alone - states from the rollout (the restoration pass, here over a horizon of more than one
    64-stage block), multipliers from the least-squares estimate - and from there follows the restatement started the same way (oracle/ipopt_ms.py: u_init)
    iteration by iteration; rocket landing, T = 100, hover-thrust guess."""
    from oracle import ipopt_ms, models, pdp_oracle as po
    from pdp_amd import zoo
    d = load(golden_dir, "demos_rocket.npz")
    st = models.IRL_SETUP["rocket"]
    oc = po.make_oc(models.REGISTRY["rocket"](**st["kwargs"]), st["dt"])
    T = 100
    rng = np.random.default_rng(11)
    u0 = np.tile(np.array([9.0, 0.0, 0.0]), (T, 1)) + 0.05 * rng.standard_normal((T, 3))
    log = []
    ref = ipopt_ms.solve(oc, d["state"][0, 0], T, d["true_parameter"], tol=1e-10, log=log, u_init=u0)
    mdl = zoo.get("rocket", "irl")
    B = 3
    sol = mdl.oc_solve_ms(np.repeat(d["state"][:1, 0], B, axis=0), d["true_parameter"], T, tol=1e-10, log_rows=len(log) + 4, u_init=np.repeat(u0[None], B, axis=0))
    assert bool(sol["converged"].all()) and (sol["iterations"].cpu().numpy() == ref["iterations"]).all() and ref["iterations"] == len(log)
    kl = sol["log"][B - 1].cpu().numpy()
    for r, l in zip(kl, log):
        assert r[5] == _alpha_col(l) and abs(r[4] - l["dw"]) <= 1e-12 * max(1.0, l["dw"]), (l["it"], r[4], r[5], l["dw"], l["alpha"])
        assert abs(r[1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(r[2] - l["inf_pr"]) <= 1e-9 * max(1.0, l["inf_pr"])
    assert log[0]["inf_pr"] <= 1e-12                                           # the starting point is the rollout: feasible
    sc = lambda a: max(1.0, np.abs(a).max())
    for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
        assert np.abs(sol[k].cpu().numpy() - ref[kr][None]).max() <= 1e-8 * sc(ref[kr])


def test_ms_kernel_warm_start_gains_and_per_sample_parameters(golden_dir):
    """256 cart-pole problems with per-sample parameters (BASELINE config C2): cold solve, then a warm start (PDP_MS_WARM) from that
    solution at perturbed parameters converges in a few iterations to a KKT point of the new problem; the gains output drives a
    descent closed-loop rollout."""
    import torch
    from pdp_amd import zoo
    d = load(golden_dir, "demos_cartpole.npz")
    mdl = zoo.get("cartpole", "irl")
    rng = np.random.default_rng(3)
    B, T = 256, 50
    x0 = np.zeros((B, 4))
    x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    th = d["true_parameter"][None] * (1 + 0.1 * rng.uniform(-1, 1, (B, 7)))
    cold = mdl.oc_solve_ms(x0, th, T, want_gains=True)
    assert int(cold["converged"].sum()) >= 0.95 * B
    ok = cold["converged"]
    th2 = th * (1 + 0.02 * rng.uniform(-1, 1, th.shape))
    warm = mdl.oc_solve_ms(x0, th2, T, warm=(cold["state"], cold["control"], cold["costate"]))
    both = ok & warm["converged"]
    assert int(both.sum()) >= 0.95 * B
    assert float(warm["iterations"][both].double().mean()) <= 8 and float(warm["iterations"][both].double().mean()) < float(cold["iterations"][both].double().mean())
    # KKT point of the new problem, checked stage by stage with the single-purpose kernels (the swing-up is unstable in open loop: a
    # T-step rollout would amplify the 1e-10 defects): x_{t+1} = f(x_t, u_t), lam the costate recursion along (x, u), H_u = 0
    xs, us = warm["state"], warm["control"]
    x1, _ = mdl.oc_rollout(xs[:, :-1].reshape(-1, 4), us.reshape(-1, 1, 1), np.repeat(th2, T, axis=0))
    defect = (x1[:, 1].reshape(B, T, 4) - xs[:, 1:]).abs().amax(dim=(1, 2))
    assert float(defect[both].max()) <= 1e-9 * (1 + float(xs[both].abs().max()))
    assert float((defect - warm["resid"][:, 0])[both].abs().max()) <= 1e-12 * (1 + float(xs[both].abs().max()))
    lam = mdl.oc_costate(xs, us, th2)
    assert float((lam - warm["costate"])[both].abs().max()) <= 1e-7 * (1 + float(lam[both].abs().max()))
    hu = mdl.oc_auxsys(xs, us, warm["costate"], th2, only=("dHu",))["dHu"]
    assert float(hu[both].abs().max()) <= 1e-8 * (1 + float(lam[both].abs().max()))
    from test_gpu_models import oracle_oc
    oc = oracle_oc("cartpole")
    for i in torch.nonzero(both).flatten()[:3].tolist():
        Ji = oc.cost(xs[i].cpu().numpy(), us[i].cpu().numpy(), th2[i])
        assert abs(Ji - float(warm["cost"][i])) <= 1e-11 * abs(Ji)
    xp = x0 + 0.02 * rng.standard_normal(x0.shape)
    _, _, J_cl = mdl.oc_rollout_feedback(xp, cold["control"], cold["state"], cold["gains"], torch.zeros((B,), dtype=torch.float64, device="cuda"), th)
    _, J_ol = mdl.oc_rollout(xp, cold["control"], th)
    assert bool((J_cl[ok] <= J_ol[ok] + 1e-9 * J_ol[ok].abs()).all())


def test_oc_solver_with_state_and_control_bounds(golden_dir):
    """OCSys.ocSolver with finite bounds (the reference passes them to IPOPT as lbw / ubw, PDP.py:141-168): pendulum swing-up with |u| <= 12 and dq <= 6,
    both active.  Known answer: tests/golden/bounded_oc_pendulum.npz, the same problem solved by an independent method on the oracle's model (scipy SLSQP,
    single shooting: make_bounded_oc.py; its own starts agree to 3e-8 in the cost).  Also: the bounds hold, the returned cost is the original objective
    along the returned trajectory, the batch entry point with per-sample parameters; an initial state outside the state bounds is refused."""
    from pdp_amd import PDP, zoo
    from pdp_amd.sx import vertcat
    g = load(golden_dir, "bounded_oc_pendulum.npz")
    env, dt = zoo.make_env("pendulum", "irl")
    umax, vmax, T = float(g["umax"]), float(g["vmax"]), int(g["T"])
    oc = PDP.OCSys("pendulum bounded")
    oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar))
    oc.setStateVariable(env.X, state_lb=[-1e20, -1e20], state_ub=[1e20, vmax])
    oc.setControlVariable(env.U, control_lb=[-umax], control_ub=[umax])
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    assert oc.has_bounds()
    traj = oc.ocSolver(ini_state=g["x0"], horizon=T, auxvar_value=g["theta"])
    x, u = traj["state_traj_opt"], traj["control_traj_opt"]
    assert np.all(np.abs(u) <= umax) and np.all(x[1:, 1] <= vmax)
    assert abs(float(np.asarray(traj["cost"]).squeeze()) - float(g["cost"])) <= 2e-7 * float(g["cost"])
    assert np.abs(u - g["control"]).max() <= 2e-3 * umax and np.abs(x - g["state"]).max() <= 2e-3 * np.abs(g["state"]).max()
    assert np.array_equal(np.abs(np.abs(u[:, 0]) - umax) < 1e-5, g["active_u"]) and np.array_equal(np.abs(x[1:, 1] - vmax) < 1e-5, g["active_v"])
    # cost is the original objective of the rollout of the returned controls
    xr, cr = oc.rollout_batch(g["x0"][None], u[None], g["theta"])
    assert abs(float(cr[0]) - float(np.asarray(traj["cost"]).squeeze())) <= 1e-12 * float(cr[0]) and float(np.abs(xr[0].cpu().numpy() - x).max()) <= 1e-6
    # batch with per-sample parameters and initial states
    B = 3
    th = g["theta"][None] * (1 + 0.05 * np.arange(B)[:, None])
    x0 = np.tile(g["x0"], (B, 1))
    x0[1, 1] = 2.0
    x0[2, 0] = 0.3
    sol = oc.ocSolver_batch(x0, T, th)
    assert bool(sol["converged"].all())
    xs, us = sol["state"].cpu().numpy(), sol["control"].cpu().numpy()
    assert np.all(np.abs(us) <= umax) and np.all(xs[:, 1:, 1] <= vmax + 1e-9)
    assert np.abs(us[0] - u).max() <= 1e-5 * umax
    x0[1, 1] = vmax + 1.0
    with pytest.raises(NotImplementedError, match="outside the state bounds"):
        oc.ocSolver_batch(x0, T, th)


def test_control_bounded_cartpole_and_an_initial_state_on_a_state_bound(golden_dir):
    """(i) cart-pole swing-up of the IRL example (T = 30) with |u| <= 10, eight control bounds active: known answer tests/golden/bounded_oc_cartpole.npz (scipy SLSQP on
    the single-shooting problem, six starts agreeing to 1e-7 in the cost: make_bounded_oc.py).  (ii) An initial state exactly ON a state bound is accepted - the
    reference's NLP does not bound x_0 (PDP.py:141-146) and IPOPT relaxes every bound by 1e-8 (bound_relax_factor), as ocsolver.relaxed_bounds does: pendulum started at
    dq_0 = -6 with -6 <= dq <= 6; the bound is inactive afterwards, so the solution equals that of the same problem with the lower bound moved away.  (iii) keyword
    arguments the barrier continuation does not serve are refused instead of being dropped."""
    from pdp_amd import PDP, zoo
    from pdp_amd.sx import vertcat
    g = load(golden_dir, "bounded_oc_cartpole.npz")
    env, dt = zoo.make_env("cartpole", "irl")
    umax, T = float(g["umax"]), int(g["T"])
    oc = PDP.OCSys("cartpole bounded")
    oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar))
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U, control_lb=[-umax], control_ub=[umax])
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    sol = oc.ocSolver_batch(g["x0"][None], T, g["theta"])
    assert bool(sol["converged"].all()) and "kernel_converged" in sol
    x, u = sol["state"][0].cpu().numpy(), sol["control"][0].cpu().numpy()
    assert np.all(np.abs(u) <= umax)
    assert abs(float(sol["cost"][0]) - float(g["cost"])) <= 1e-6 * float(g["cost"])
    assert np.abs(u - g["control"]).max() <= 5e-3 * umax and np.abs(x - g["state"]).max() <= 5e-3 * np.abs(g["state"]).max()
    assert np.array_equal(np.abs(np.abs(u[:, 0]) - umax) < 1e-5, g["active_u"])
    with pytest.raises(NotImplementedError, match="warm_start"):
        oc.ocSolver_batch(g["x0"][None], T, g["theta"], warm_start=sol)
    # (ii)
    gp = load(golden_dir, "bounded_oc_pendulum.npz")
    envp, dtp = zoo.make_env("pendulum", "irl")

    def pend(lb):
        o = PDP.OCSys("pendulum bounded x0")
        o.setAuxvarVariable(vertcat(envp.dyn_auxvar, envp.cost_auxvar))
        o.setStateVariable(envp.X, state_lb=[-1e20, lb], state_ub=[1e20, 6.0])
        o.setControlVariable(envp.U, control_lb=[-12.0], control_ub=[12.0])
        o.setDyn(envp.X + dtp * envp.f)
        o.setPathCost(envp.path_cost)
        o.setFinalCost(envp.final_cost)
        return o
    x0 = np.array([[0.0, -6.0]])
    on = pend(-6.0).ocSolver_batch(x0, 20, gp["theta"])
    off = pend(-9.0).ocSolver_batch(x0, 20, gp["theta"])
    assert bool(on["converged"].all()) and bool(off["converged"].all())
    xs = on["state"][0].cpu().numpy()
    assert xs[0, 1] == -6.0 and np.all(xs[1:, 1] > -6.0 + 1e-3) and np.all(xs[1:, 1] <= 6.0 + 1e-9)       # on the bound at t = 0, inside afterwards
    assert float((on["state"] - off["state"]).abs().max()) <= 1e-5 * float(off["state"].abs().max())
    assert abs(float(on["cost"][0]) - float(off["cost"][0])) <= 1e-7 * float(off["cost"][0])
