"""CPU: pin the oracle (oracle/pdp_oracle.py) against
  (1) the reference's stored CasADi+IPOPT results (tests/golden/{demos,iodata,irltrace,oc_*}.npz), and
  (2) outputs of the reference's own PDP.py executed in the build container (tests/golden/ref_*.npz).
Tolerances: forward/first-order quantities 1e-12; full IRL pipeline loss 1e-7 rel / gradient 1e-6 rel
(limited by IPOPT's 1e-8 tolerance and the (p_k - p_{k+1})/lr quantisation of the stored trace)."""
import os

import numpy as np
import pytest

from oracle import models, pdp_oracle as po

SYSTEMS = ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"]
_cache = {}


def _oc(name):
    if ("oc", name) not in _cache:
        st = models.IRL_SETUP[name]
        _cache[("oc", name)] = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    return _cache[("oc", name)]


def _sysid(name):
    if ("id", name) not in _cache:
        st = models.SYSID_SETUP[name]
        _cache[("id", name)] = po.make_sysid(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    return _cache[("id", name)]


def _load(golden_dir, f):
    return np.load(os.path.join(golden_dir, f))


@pytest.mark.parametrize("name", SYSTEMS)
def test_forward_integrator_matches_stored_iodata(golden_dir, name):
    io = _load(golden_dir, "iodata_%s.npz" % name)
    sid = _sysid(name)
    for i in range(io["inputs"].shape[0]):
        xs = sid.integrateDyn(io["states"][i, 0], io["inputs"][i], io["true_parameter"])
        scale = max(1.0, np.abs(io["states"][i]).max())
        assert np.abs(xs - io["states"][i]).max() <= 1e-13 * scale
    loss, grad = sid.step(list(io["inputs"]), list(io["states"]), io["true_parameter"])
    assert loss < 1e-24


@pytest.mark.parametrize("name", SYSTEMS)
def test_pmp_conditions_hold_on_stored_demos(golden_dir, name):
    """costate convention costate[t] = lambda_{t+1} (IPOPT lam_g sign), dH/du = 0, costate recursion, cost."""
    d = _load(golden_dir, "demos_%s.npz" % name)
    oc = _oc(name)
    th = d["true_parameter"]
    for i in range(d["state"].shape[0]):
        xs, us, ls = d["state"][i], d["control"][i], d["costate"][i]
        ru, rx, rc = oc.kkt_residual(xs, us, ls, th)
        assert np.abs(ru).max() < 1e-8 and np.abs(rx).max() < 1e-7 and np.abs(rc).max() < 1e-8
        assert abs(oc.cost(xs, us, th) - d["cost"][i]) <= 1e-11 * abs(d["cost"][i])
        assert np.abs(oc.rollout(xs[0], us, th) - xs).max() < 1e-6
        assert np.abs(oc.costate(xs, us, th) - ls).max() <= 1e-6 * max(1.0, np.abs(ls).max())


@pytest.mark.parametrize("name,rows", [("pendulum", [0, 3, 9]), ("cartpole", [0, 3, 7]), ("robotarm", [0, 4]),
                                       ("quadrotor", [1, 5, 9]), ("rocket", [0, 3, 8])])
def test_full_irl_pipeline_matches_stored_trace(golden_dir, margins, name, rows):
    """OC solve -> getAuxSys -> lqrSolver -> chain rule reproduces loss_trace[k+1] and (p_k-p_{k+1})/lr."""
    d = _load(golden_dir, "demos_%s.npz" % name)
    tr = _load(golden_dir, "irltrace_%s.npz" % name)
    oc = _oc(name)
    T = d["control"].shape[1]
    for j in rows:
        th = tr["param"][j]
        loss, dp = 0.0, np.zeros(oc.p)
        for i in range(d["state"].shape[0]):
            sol = po.solve_oc_homotopy(oc, d["state"][i, 0], T, th, d["true_parameter"], (d["state"][i], d["control"][i], d["costate"][i]))
            aux = oc.getAuxSys(sol["state_traj_opt"], sol["control_traj_opt"], sol["costate_traj_opt"], th)
            lq = po.lqr_from_aux(aux, oc.n, oc.p, T)
            l, g = po.irl_loss_grad(sol["state_traj_opt"], sol["control_traj_opt"], d["state"][i], d["control"][i],
                                    lq["state_traj_opt"], lq["control_traj_opt"])
            loss += l
            dp += g
        nd = d["state"].shape[0]
        loss, dp = loss / nd, dp / nd
        gref = (tr["param"][j] - tr["param_next"][j]) / float(tr["lr"])
        # restatement against the stored traces <= 1e-9 (loss), <= 1e-7 (gradient)
        margins.check("oracle vs stored IRL trace, %s row %d: loss (relative)" % (name, j), abs(loss - tr["loss_next"][j]) / abs(tr["loss_next"][j]), 1e-9)
        margins.check("oracle vs stored IRL trace, %s row %d: gradient (relative to its largest entry)" % (name, j), np.abs(dp - gref).max() / np.abs(gref).max(), 1e-7)


def test_quadrotor_oc_stored_rollout_and_cost(golden_dir):
    """OC/quadrotor stored solution: Lagrange policy rollout + cost incl. the attitude term."""
    g = _load(golden_dir, "oc_quadrotor.npz")
    m = models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1)
    cp = po.make_cp(m, float(g["dt"]))
    T = int(g["horizon"])
    cp.init_step(T)
    B = np.stack([cp._basis(t) for t in range(T)])
    theta, res, *_ = np.linalg.lstsq(B, g["control"], rcond=None)      # recover the 6 pivots from the stored controls
    sol = cp.integrateSys(g["state"][0], T, theta.reshape(-1))
    assert np.abs(sol["state_traj"] - g["state"]).max() < 1e-11
    assert abs(sol["cost"] - float(g["cost"])) < 1e-9 * float(g["cost"])
    # the stored IPOPT optimum must be cheaper than the PDP policy solution
    assert float(g["true_cost"]) < float(g["cost"])


def test_mlp_parameter_layout_is_column_major(golden_dir):
    g = _load(golden_dir, "oc_cartpole_neural.npz")
    m = models.cart_pole(mc=0.1, mp=0.1, l=1, wx=0.1, wq=0.6, wdx=0.1, wdq=0.1, wu=0.3)   # OC/cartpole/cartpole_PDP_neural.py
    cp = po.make_cp(m, float(g["dt"]))
    cp.init_step_neural_policy([4, 4])
    T = g["control"].shape[0]
    sol = cp.integrateSys(g["state"][0], T, g["param_final"])
    assert np.abs(sol["state_traj"] - g["state"]).max() < 1e-10
    assert np.abs(sol["control_traj"] - g["control"]).max() < 1e-10


# ---------------------- outputs of the reference's own PDP.py (make_ref_outputs.py) -----------------------
@pytest.mark.parametrize("name", SYSTEMS)
def test_getAuxSys_and_lqrSolver_match_reference_run(golden_dir, name):
    d = _load(golden_dir, "demos_%s.npz" % name)
    ra = _load(golden_dir, "ref_auxsys_%s.npz" % name)
    rl = _load(golden_dir, "ref_lqr_%s.npz" % name)
    oc = _oc(name)
    T = d["control"].shape[1]
    for i in range(d["state"].shape[0]):
        aux = oc.getAuxSys(d["state"][i], d["control"][i], d["costate"][i], ra["theta"])
        for k in ["dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue", "hxx", "hxe"]:
            ref = ra[k][i]
            assert np.abs(np.stack(aux[k]) - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), k
        sol = po.lqr_from_aux(aux, oc.n, oc.p, T)
        for k, kk in [("X", "state_traj_opt"), ("U", "control_traj_opt"), ("Lam", "costate_traj_opt")]:
            ref = rl[k][i]
            assert np.abs(np.stack(sol[kk]) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), k


def test_lqrSolver_random_cases_match_reference_run(golden_dir):
    r = _load(golden_dir, "ref_lqr_random.npz")
    for c in range(int(r["n_cases"])):
        g = lambda k: r["c%d_%s" % (c, k)]
        T = int(g("T"))
        bc = (lambda a: [a[t] for t in range(T)]) if bool(g("time_varying")) else (lambda a: T * [a[0]])
        sol = po.lqr_solver(bc(g("F")), bc(g("G")), bc(g("E")), bc(g("Hxx")), bc(g("Huu")), bc(g("Hxu")), bc(g("Hxe")), bc(g("Hue")),
                            [g("hxx")], [g("hxe")], g("X0"), T)
        assert np.abs(np.stack(sol["state_traj_opt"]) - g("X")).max() < 1e-12 * max(1, np.abs(g("X")).max())
        assert np.abs(np.stack(sol["control_traj_opt"]) - g("U")).max() < 1e-12 * max(1, np.abs(g("U")).max())
        assert np.abs(np.stack(sol["costate_traj_opt"]) - g("Lam")).max() < 1e-12 * max(1, np.abs(g("Lam")).max())


CP_CASES = {
    "pendulum_poly": lambda: models.single_pendulum(l=1, m=1, damping_ratio=0.05, wq=10, wdq=1, wu=0.1),
    "cartpole_mlp": lambda: models.cart_pole(mc=0.1, mp=0.1, l=1, wx=0.1, wq=0.6, wdx=0.1, wdq=0.1, wu=0.3),
    "robotarm_mlp": lambda: models.robot_arm(l1=1, m1=1, l2=1, m2=1, g=0, wq1=0.1, wq2=0.1, wdq1=0.1, wdq2=0.1, wu=0.01),
    "quadrotor_poly": lambda: models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1),
    "quadrotor_mlp": lambda: models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1),
    "rocket_poly": lambda: models.rocket(Jx=0.5, Jy=1, Jz=1, mass=1, l=1, wr=1, wv=1, wtilt=50, ww=1, wsidethrust=1, wthrust=0.4),
}


def make_cp_case(tag, g):
    cp = po.make_cp(CP_CASES[tag](), float(g["dt"]))
    if tag.endswith("poly"):
        cp.init_step(int(g["T"]))
    else:
        h = [int(v) for v in g["hidden"]]
        cp.init_step_neural_policy(h if h else None)
    return cp


@pytest.mark.parametrize("tag", sorted(CP_CASES))
def test_control_planning_step_matches_reference_run(golden_dir, tag):
    g = _load(golden_dir, "ref_cp_%s.npz" % tag)
    cp = make_cp_case(tag, g)
    T = int(g["T"])
    assert cp.n_auxvar == g["theta"].size
    sol = cp.integrateSys(g["x0"], T, g["theta"])
    assert np.abs(sol["state_traj"] - g["state"]).max() <= 1e-11 * max(1, np.abs(g["state"]).max())
    aux = cp.getAuxSys(sol["state_traj"], sol["control_traj"], g["theta"])
    for k in ["dynF", "dynG", "dUx", "dUe"]:
        assert np.abs(np.stack(aux[k]) - g[k]).max() <= 1e-11 * max(1, np.abs(g[k]).max()), k
    loss, grad = cp.step(g["x0"], T, g["theta"])
    assert abs(loss - float(g["loss"])) <= 1e-11 * abs(float(g["loss"]))
    assert np.abs(grad - g["grad"]).max() <= 1e-10 * np.abs(g["grad"]).max()


@pytest.mark.parametrize("name", SYSTEMS)
def test_sysid_step_matches_reference_run(golden_dir, name):
    g = _load(golden_dir, "ref_sysid_%s.npz" % name)
    io = _load(golden_dir, "iodata_%s.npz" % name)
    loss, grad = _sysid(name).step(list(io["inputs"]), list(io["states"]), g["theta"])
    assert abs(loss - float(g["loss"])) <= 1e-12 * abs(float(g["loss"]))
    assert np.abs(grad - g["grad"]).max() <= 1e-11 * np.abs(g["grad"]).max()


@pytest.mark.parametrize("name,demos", [("pendulum", [0, 1, 2, 3, 4]), ("robotarm", [0, 1, 2]), ("rocket", [0]), ("quadrotor", [0]), ("cartpole", [3])])
def test_ipopt_restatement_reaches_the_stored_optimum_from_the_zero_guess(golden_dir, name, demos):
    """oracle/ipopt_ms.py (IPOPT's published algorithm on the reference's multiple-shooting NLP, PDP.py:131-182, all-zero initial
    guess) is pinned on what the real IPOPT returned on the author's machine: state, control, lam_g and cost of the stored demos."""
    from oracle import ipopt_ms
    d = _load(golden_dir, "demos_%s.npz" % name)
    oc = _oc(name)
    T = d["control"].shape[1]
    for i in demos:
        s = ipopt_ms.solve(oc, d["state"][i, 0], T, d["true_parameter"])
        sc = lambda a: max(1.0, np.abs(a).max())
        assert abs(s["cost"] - d["cost"][i]) <= 1e-9 * abs(d["cost"][i])
        assert np.abs(s["state_traj_opt"] - d["state"][i]).max() <= 1e-7 * sc(d["state"][i])
        assert np.abs(s["control_traj_opt"] - d["control"][i]).max() <= 1e-7 * sc(d["control"][i])
        assert np.abs(s["costate_traj_opt"] - d["costate"][i]).max() <= 1e-7 * sc(d["costate"][i])


def test_ipopt_restatement_reproduces_the_stored_rocket_irl_trace(golden_dir):
    """the reference's IRL loop solves the OC problem cold at every iterate (Examples/IRL/rocket/rocket_PDP.py); the restatement lands
    in the same optimum at iterates far from theta* (loss 980) and near it: stored loss_trace reproduced to 1e-9 relative."""
    from oracle import ipopt_ms
    d = _load(golden_dir, "demos_rocket.npz")
    tr = _load(golden_dir, "irltrace_rocket.npz")
    oc = _oc("rocket")
    T = d["control"].shape[1]
    for j in (0, 4):
        s = ipopt_ms.solve(oc, d["state"][0, 0], T, tr["param"][j])
        loss = ((s["state_traj_opt"] - d["state"][0]) ** 2).sum() + ((s["control_traj_opt"] - d["control"][0]) ** 2).sum()
        assert abs(loss - tr["loss_next"][j]) <= 1e-9 * abs(tr["loss_next"][j])


def test_ipopt_restatement_restoration_phase(golden_dir):
    """robot arm demo 3 starts at an equilibrium of the zero guess: the line search falls below alpha_min after 8 iterations, where IPOPT enters its
    restoration phase.  The restatement's restoration (states <- rollout of the current controls, current point into the filter, multipliers from the
    least-squares estimate; oracle/ipopt_ms.py: solve) continues from there and lands in the optimum the real IPOPT stored for this demo; switched
    off, the restatement reports the point where it would have been needed."""
    from oracle import ipopt_ms
    d = _load(golden_dir, "demos_robotarm.npz")
    oc, T = _oc("robotarm"), d["control"].shape[1]
    with pytest.raises(RuntimeError, match="restoration"):
        ipopt_ms.solve(oc, d["state"][3, 0], T, d["true_parameter"], restoration=False)
    log = []
    s = ipopt_ms.solve(oc, d["state"][3, 0], T, d["true_parameter"], log=log)
    assert s["restorations"] == 1 and [bool(l.get("restoration")) for l in log].index(True) == 8
    assert abs(s["cost"] - d["cost"][3]) <= 1e-12 * abs(d["cost"][3])
    assert np.abs(s["state_traj_opt"] - d["state"][3]).max() <= 1e-9 and np.abs(s["control_traj_opt"] - d["control"][3]).max() <= 1e-9
    assert np.abs(s["costate_traj_opt"] - d["costate"][3]).max() <= 1e-9 * max(1.0, np.abs(d["costate"][3]).max())
    # the demos that never get there are untouched by the switch
    s0 = ipopt_ms.solve(oc, d["state"][0, 0], T, d["true_parameter"])
    assert s0["restorations"] == 0

