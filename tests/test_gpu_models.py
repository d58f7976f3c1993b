"""GPU parity of the generated-model kernels (through the C-ABI of libpdp_model_*.so) against the oracle, the
reference's stored CasADi/IPOPT results and the outputs of the reference's own PDP.py (tests/golden).
Stated fp64 tolerance: 1e-10 relative to the largest entry of the compared array (TOL); aux matrices 1e-11;
1e-6 (gradient) / 1e-7 (loss) against the stored IPOPT traces (IPOPT's own 1e-8 tolerance).
Where the reference's order of operations is itself ill-conditioned (inverse of I + P R on off-optimal
trajectories, PDP.py:575) parity is judged against the same formulas in 40-digit arithmetic
(oracle.lqr_solver_mp) and the fp64 reference order is only required to agree to its own measured accuracy."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SYSTEMS = ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"]
TOL = 1e-10
_cache = {}


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def npy(t):
    return t.detach().cpu().numpy()


def oracle_oc(name):
    from oracle import models, pdp_oracle as po
    if ("oc", name) not in _cache:
        st = models.IRL_SETUP[name]
        _cache[("oc", name)] = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    return _cache[("oc", name)]


def oracle_sysid(name):
    from oracle import models, pdp_oracle as po
    if ("id", name) not in _cache:
        st = models.SYSID_SETUP[name]
        _cache[("id", name)] = po.make_sysid(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    return _cache[("id", name)]


def load(golden_dir, f):
    return np.load(os.path.join(golden_dir, f))


# ------------------------------------------------------------------------------------------------ OC / IRL
@pytest.mark.parametrize("name", SYSTEMS)
def test_oc_rollout_costate_auxsys(golden_dir, name):
    from pdp_amd import zoo
    mdl = zoo.get(name, "irl")
    d = load(golden_dir, "demos_%s.npz" % name)
    ra = load(golden_dir, "ref_auxsys_%s.npz" % name)
    oc = oracle_oc(name)
    th = d["true_parameter"]
    x, cost = mdl.oc_rollout(d["state"][:, 0], d["control"], th)
    lam = mdl.oc_costate(x, d["control"], th)
    x, cost, lam = npy(x), npy(cost), npy(lam)
    for i in range(d["state"].shape[0]):
        xo = oc.rollout(d["state"][i, 0], d["control"][i], th)
        assert rel(x[i], xo) < TOL                 # unstable open-loop dynamics amplify 1-ulp differences of sin/cos
        assert abs(cost[i] - oc.cost(xo, d["control"][i], th)) < TOL * abs(cost[i])
        assert rel(lam[i], oc.costate(x[i], d["control"][i], th)) < TOL
    # aux system at the stored optimum vs the reference's own getAuxSys
    aux = mdl.oc_auxsys(d["state"], d["control"], d["costate"], th)
    for k in ["dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue"]:
        assert rel(npy(aux[k]), ra[k]) < 1e-11, k
    assert rel(npy(aux["hxx"]), ra["hxx"][:, 0]) < 1e-11 and rel(npy(aux["hxe"]), ra["hxe"][:, 0]) < 1e-11


@pytest.mark.parametrize("name", SYSTEMS)
def test_fused_pdp_given_optimal_trajectory_matches_reference_lqr(golden_dir, name):
    """PDP_OC_GIVEN_TRAJ on the stored demos: dx/dtheta, du/dtheta equal the reference lqrSolver output."""
    from pdp_amd import zoo
    mdl = zoo.get(name, "irl")
    d = load(golden_dir, "demos_%s.npz" % name)
    rl = load(golden_dir, "ref_lqr_%s.npz" % name)
    rng = np.random.default_rng(3)
    demo_x = d["state"] + 0.1 * rng.standard_normal(d["state"].shape)
    demo_u = d["control"] + 0.1 * rng.standard_normal(d["control"].shape)
    out = mdl.oc_pdp_grad(d["control"], d["true_parameter"], demo_x, demo_u, x=d["state"], lam=d["costate"], want_sens=True)
    assert int(out["status"].sum()) == 0
    assert rel(npy(out["dxdp"]), rl["X"]) < 1e-10
    assert rel(npy(out["dudp"]), rl["U"]) < 1e-10
    from oracle import pdp_oracle as po
    for i in range(d["state"].shape[0]):
        l, g = po.irl_loss_grad(d["state"][i], d["control"][i], demo_x[i], demo_u[i], list(rl["X"][i]), list(rl["U"][i]))
        assert abs(npy(out["loss"])[i] - l) < 1e-12 * l
        assert rel(npy(out["grad"])[i], g) < 1e-10


@pytest.mark.parametrize("name", SYSTEMS)
def test_fused_pdp_unit_matches_oracle(golden_dir, name):
    """IPOPT-free unit: rollout of perturbed demo controls at a perturbed theta -> costates -> aux -> Riccati -> gradient."""
    from oracle import pdp_oracle as po
    from pdp_amd import zoo
    mdl = zoo.get(name, "irl")
    d = load(golden_dir, "demos_%s.npz" % name)
    oc = oracle_oc(name)
    rng = np.random.default_rng(7)
    nd = d["state"].shape[0]
    B = 2 * nd
    u = np.concatenate([d["control"], d["control"]]) * (1 + 0.02 * rng.standard_normal((B,) + d["control"].shape[1:]))
    x0 = np.concatenate([d["state"][:, 0], d["state"][:, 0]])
    theta = d["true_parameter"][None, :] * (1 + 0.05 * rng.standard_normal((B, d["true_parameter"].size)))
    demo_x = np.concatenate([d["state"], d["state"]])
    demo_u = np.concatenate([d["control"], d["control"]])
    out = mdl.oc_pdp_grad(u, theta, demo_x, demo_u, x0=x0, want_sens=True)
    assert int(out["status"].sum()) == 0
    xg, lg = npy(out["x"]), npy(out["lam"])
    for i in range(B):
        xs = oc.rollout(x0[i], u[i], theta[i])
        assert rel(xg[i], xs) < TOL
        assert rel(lg[i], oc.costate(xg[i], u[i], theta[i])) < TOL
        # aux system + reference formulas on the kernel's own trajectory: fp64 reference order and 40-digit evaluation
        aux = oc.getAuxSys(xg[i], u[i], lg[i], theta[i])
        T = u.shape[1]
        ref64 = po.lqr_from_aux(aux, oc.n, oc.p, T)
        # every sample is held to the 40-digit evaluation of the reference formulas (no tolerance constants per sample)
        ex = po.lqr_solver_mp(aux["dynF"], aux["dynG"], aux["dynE"], aux["Hxx"], aux["Huu"], aux["Hxu"], aux["Hxe"], aux["Hue"],
                              aux["hxx"], aux["hxe"], np.zeros((oc.n, oc.p)), T)
        Xe, Ue = np.stack(ex["state_traj_opt"]), np.stack(ex["control_traj_opt"])
        ref_err = rel(np.stack(ref64["state_traj_opt"]), Xe)          # the fp64 reference order's own rounding error on this sample
        tol_i = max(TOL, 2 * ref_err)                                 # ill-conditioned samples: at least as accurate as the reference order
        assert rel(npy(out["dxdp"])[i], Xe) < tol_i and rel(npy(out["dudp"])[i], Ue) < tol_i
        l, g = po.irl_loss_grad(xg[i], u[i], demo_x[i], demo_u[i], list(Xe), list(Ue))
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < tol_i
        assert rel(npy(out["dxdp"])[i], np.stack(ref64["state_traj_opt"])) < max(TOL, 10 * ref_err)


@pytest.mark.parametrize("name,rows", [("cartpole", [0, 3, 7]), ("quadrotor", [1, 5, 9]), ("rocket", [0, 3, 8]), ("pendulum", [0, 9]), ("robotarm", [0, 4])])
def test_full_irl_iteration_matches_stored_reference_trace(golden_dir, margins, name, rows):
    """The reference's stored (parameter_trace, loss_trace): optimum at theta_k (oracle Newton-KKT stands in for IPOPT)
    -> HIP fused aux+Riccati+gradient reproduces loss_trace[k+1] and (p_k - p_{k+1})/lr."""
    from oracle import pdp_oracle as po
    from pdp_amd import zoo
    mdl = zoo.get(name, "irl")
    d = load(golden_dir, "demos_%s.npz" % name)
    tr = load(golden_dir, "irltrace_%s.npz" % name)
    oc = oracle_oc(name)
    T = d["control"].shape[1]
    for j in rows:
        th = tr["param"][j]
        sols = [po.solve_oc_homotopy(oc, d["state"][i, 0], T, th, d["true_parameter"], (d["state"][i], d["control"][i], d["costate"][i]))
                for i in range(d["state"].shape[0])]
        xs = np.stack([s["state_traj_opt"] for s in sols])
        us = np.stack([s["control_traj_opt"] for s in sols])
        ls = np.stack([s["costate_traj_opt"] for s in sols])
        out = mdl.oc_pdp_grad(us, th, d["state"], d["control"], x=xs, lam=ls)
        loss = float(npy(out["loss"]).mean())
        dp = npy(out["grad"]).mean(axis=0)
        gref = (tr["param"][j] - tr["param_next"][j]) / float(tr["lr"])
        # <= 1e-9 (loss), <= 1e-7 (gradient) against the stored traces
        margins.check("GPU gradient unit at the oracle's optimum vs stored IRL trace, %s row %d: loss (relative)" % (name, j),
                      abs(loss - tr["loss_next"][j]) / abs(tr["loss_next"][j]), 1e-9)
        margins.check("GPU gradient unit at the oracle's optimum vs stored IRL trace, %s row %d: gradient (relative to its largest entry)" % (name, j),
                      np.abs(dp - gref).max() / np.abs(gref).max(), 1e-7)


def test_fused_pdp_full_size_batch_properties():
    """BASELINE config C3 size (quadrotor n=13 T=50 B=1024): per-sample results do not depend on the batch they are in, shared theta ==
    replicated theta, status clean.  Batches that run in the same workgroup shape (the library picks 1, 2 or 4 trajectories per workgroup by
    batch size, pdp_model.hip) agree BIT FOR BIT; across shapes the instantiations of the kernel may contract a few products differently:
    agreement to 1e-13."""
    from pdp_amd import zoo
    import torch
    mdl = zoo.get("quadrotor", "irl")
    rng = np.random.default_rng(0)
    B, T = 1024, 50
    x0 = np.zeros((B, 13)); x0[:, :2] = rng.uniform(-8, 8, (B, 2)); x0[:, 2] = rng.uniform(5, 10, B); x0[:, 6] = 1.0
    u = 2.5 + 0.3 * rng.standard_normal((B, T, 4))
    th = np.array([1, 1, 1, 1, .4, 1, 1, 5, 1.0])
    dx = np.zeros((B, T + 1, 13)); dx[:, :, 6] = 1.0
    du = np.full((B, T, 4), 2.5)
    o1 = mdl.oc_pdp_grad(u, th, dx, du, x0=x0)
    g1, l1 = npy(o1["grad"]).copy(), npy(o1["loss"]).copy()
    assert int(o1["status"].sum()) == 0 and np.all(np.isfinite(g1))
    o2 = mdl.oc_pdp_grad(u[:777], np.tile(th, (777, 1)), dx[:777], du[:777], x0=x0[:777])          # same shape as the full batch (4 per workgroup)
    assert np.array_equal(npy(o2["grad"]), g1[:777]) and np.array_equal(npy(o2["loss"]), l1[:777])
    o3 = mdl.oc_pdp_grad(u[:7], np.tile(th, (7, 1)), dx[:7], du[:7], x0=x0[:7])                  # one trajectory per workgroup
    assert np.abs(npy(o3["grad"]) - g1[:7]).max() <= 1e-13 * np.abs(g1[:7]).max() and np.abs(npy(o3["loss"]) - l1[:7]).max() <= 1e-13 * np.abs(l1[:7]).max()


# ------------------------------------------------------------------------------------------------ ControlPlanning
CP_TAGS = ["pendulum_poly", "cartpole_mlp", "robotarm_mlp", "quadrotor_poly", "quadrotor_mlp", "rocket_poly"]


def _cp_setup(g, tag):
    from pdp_amd import runtime as rt, zoo
    name, kind = tag.split("_")
    mdl = zoo.get(name, "oc")
    T = int(g["T"])
    if kind == "poly":
        pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    else:
        h = [int(v) for v in g["hidden"]] or [mdl.n]
        pol = rt.make_policy("mlp", layers=h + [mdl.m])
    return mdl, pol, T


@pytest.mark.parametrize("tag", CP_TAGS)
def test_control_planning_matches_reference_run(golden_dir, tag):
    g = load(golden_dir, "ref_cp_%s.npz" % tag)
    mdl, pol, T = _cp_setup(g, tag)
    p = g["theta"].size
    x, u, cost = mdl.cp_integrate_T(pol, p, g["x0"], g["theta"], T)
    assert rel(npy(x)[0], g["state"]) < 1e-11 and rel(npy(u)[0], g["control"]) < 1e-11
    assert abs(float(cost[0]) - float(g["loss"])) < 1e-11 * abs(float(g["loss"]))
    aux = mdl.cp_auxsys(pol, p, x, u, g["theta"])
    for k in ["dynF", "dynG", "dUx", "dUe"]:
        assert rel(npy(aux[k])[0], g[k]) < 1e-11, k
    loss, grad = mdl.cp_step(pol, p, g["x0"], g["theta"], T)
    assert abs(float(loss[0]) - float(g["loss"])) < 1e-11 * abs(float(g["loss"]))
    assert rel(npy(grad)[0], g["grad"]) < 1e-10


def test_adjoint_and_materialised_forward_mode_step_agree(golden_dir):
    """the fused adjoint kernel (MLP policy) and the reference's materialised forward-sensitivity route give the same gradient"""
    g = load(golden_dir, "ref_cp_quadrotor_mlp.npz")
    mdl, pol, T = _cp_setup(g, "quadrotor_mlp")
    p = g["theta"].size
    rng = np.random.default_rng(5)
    x0 = np.tile(g["x0"], (6, 1)) + 0.1 * rng.standard_normal((6, 13))
    theta = g["theta"][None] + 0.05 * rng.standard_normal((6, p))
    l1, g1 = mdl.cp_step(pol, p, x0, theta, T)
    l2, g2 = mdl.cp_step_materialised(pol, p, x0, theta, T)
    assert rel(npy(l1), npy(l2)) < 1e-12 and rel(npy(g1), npy(g2)) < TOL
    assert rel(npy(g1)[0:1], npy(mdl.cp_step(pol, p, x0[0:1], theta[0:1], T)[1])) == 0.0


def test_control_planning_batch_matches_oracle():
    from oracle import models, pdp_oracle as po
    from pdp_amd import runtime as rt, zoo
    mdl = zoo.get("quadrotor", "oc")
    cp = po.make_cp(models.quadrotor(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01, wr=1, wv=1, wq=5, ww=1, wthrust=0.1), 0.1)
    T, B = 50, 6
    cp.init_step(T)
    rng = np.random.default_rng(2)
    x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-5, 5, (B, 3)); x0[:, 6] = 1
    theta = rng.standard_normal((B, 24))
    pol = rt.make_policy("poly", pivots=np.linspace(0, T, 6))
    loss, grad, x, u = mdl.cp_step(pol, 24, x0, theta, T, want_traj=True)
    for i in range(B):
        l, g = cp.step(x0[i], T, theta[i])
        assert abs(float(loss[i]) - l) < 1e-11 * abs(l) and rel(npy(grad)[i], g) < 1e-10


# ------------------------------------------------------------------------------------------------ SysID
@pytest.mark.parametrize("name", SYSTEMS)
def test_sysid_matches_reference_run_and_oracle(golden_dir, name):
    from pdp_amd import zoo
    mdl = zoo.get(name, "sysid")
    g = load(golden_dir, "ref_sysid_%s.npz" % name)
    io = load(golden_dir, "iodata_%s.npz" % name)
    loss, grad = mdl.sysid_step(io["inputs"], io["states"], g["theta"])
    assert abs(float(loss.mean()) - float(g["loss"])) < 1e-11 * abs(float(g["loss"]))       # mean over the batch: PDP.py:1293-1294
    assert rel(npy(grad).mean(axis=0), g["grad"]) < 1e-10
    sid = oracle_sysid(name)
    x = mdl.sysid_integrate(io["states"][:, 0], io["inputs"], io["true_parameter"])
    assert rel(npy(x), io["states"]) < 1e-12                                                # stored CasADi states
    F, E = mdl.sysid_auxsys(x, io["inputs"], g["theta"])
    aux = sid.getAuxSys(npy(x)[0], io["inputs"][0], g["theta"])
    assert rel(npy(F)[0], np.stack(aux["dynF"])) < 1e-11 and rel(npy(E)[0], np.stack(aux["dynE"])) < 1e-11
