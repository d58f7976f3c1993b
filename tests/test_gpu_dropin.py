"""GPU: the drop-in class surface (pdp_amd.PDP / pdp_amd.JinEnv / pdp_amd.sx) used the way the reference's example
scripts use PDP / JinEnv / casadi - same calls, same return keys - checked against the reference's stored data and the
outputs of the reference's own run (tests/golden)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def npy(t):
    return t.detach().cpu().numpy()


def load(golden_dir, f):
    return np.load(os.path.join(golden_dir, f))


def test_sysid_script_flow_quadrotor(golden_dir):
    """Examples/SysID/quadrotor/uav_PDP.py:9-20, 43-45 with the stored iodata"""
    from pdp_amd import PDP, JinEnv
    uav = JinEnv.Quadrotor()
    uav.initDyn(c=0.01)
    dt = 0.1
    uavid = PDP.SysID()
    uavid.setAuxvarVariable(uav.dyn_auxvar)
    uavid.setStateVariable(uav.X)
    uavid.setControlVariable(uav.U)
    dyn = uav.X + dt * uav.f
    uavid.setDyn(dyn)
    io = load(golden_dir, "iodata_quadrotor.npz")
    g = load(golden_dir, "ref_sysid_quadrotor.npz")
    batch_inputs = [io["inputs"][i] for i in range(io["inputs"].shape[0])]
    batch_states = [io["states"][i] for i in range(io["states"].shape[0])]
    loss, dp = uavid.step(batch_inputs, batch_states, g["theta"])
    assert abs(loss - float(g["loss"])) < 1e-11 * float(g["loss"]) and rel(dp, g["grad"]) < TOL
    states = uavid.integrateDyn(auxvar_value=io["true_parameter"], ini_state=batch_states[0][0], inputs=batch_inputs[0])
    assert rel(states, batch_states[0]) < 1e-12
    aux = uavid.getAuxSys(states, batch_inputs[0], io["true_parameter"])
    sol = uavid.integrateAuxSys(aux["dynF"], aux["dynE"], np.zeros((uavid.n_state, uavid.n_auxvar)))
    assert len(aux["dynF"]) == 10 and sol["state_traj"][-1].shape == (13, 5)
    # a few gradient-descent iterations decrease the loss (the loop of uav_PDP.py:43-48)
    theta, l0 = g["theta"].copy(), loss
    for _ in range(5):
        l, d = uavid.step(batch_inputs, batch_states, theta)
        theta = theta - 1e-4 * d
    assert l < l0


def test_oc_script_flow_quadrotor(golden_dir):
    """Examples/OC/quadrotor/uav_PDP.py:9-31, 58-63 ; stored solved_solution"""
    from pdp_amd import PDP, JinEnv
    uav = JinEnv.Quadrotor()
    uav.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=0.4, c=0.01)
    uav.initCost(wr=1, wv=1, wq=5, ww=1, wthrust=0.1)
    dt = 0.1
    uavoc = PDP.ControlPlanning()
    uavoc.setStateVariable(uav.X)
    uavoc.setControlVariable(uav.U)
    dyn = uav.X + dt * uav.f
    uavoc.setDyn(dyn)
    uavoc.setPathCost(uav.path_cost)
    uavoc.setFinalCost(uav.final_cost)
    horizon = 35
    ini_state = [-8, -6, 9.] + [0.0, 0.0, 0.0] + JinEnv.toQuaternion(0, [1, -1, 1]) + [0.0, 0.0, 0.0]
    uavoc.init_step(horizon)
    g = load(golden_dir, "ref_cp_quadrotor_poly.npz")
    loss, dp = uavoc.step(ini_state, horizon, g["theta"])
    assert abs(loss - float(g["loss"])) < 1e-11 * float(g["loss"]) and rel(dp, g["grad"]) < TOL
    sol = uavoc.integrateSys(ini_state, horizon, g["theta"])
    assert set(sol) == {"state_traj", "control_traj", "cost"} and rel(sol["state_traj"], g["state"]) < 1e-11
    # stored PDP solution of the reference: its Lagrange pivots are recoverable from the stored controls
    st = load(golden_dir, "oc_quadrotor.npz")
    B = np.stack([[np.prod([(t - pj) / (pi - pj) for pj in uavoc.pivots if pj != pi]) for pi in uavoc.pivots] for t in range(horizon)])
    theta = np.linalg.lstsq(B, st["control"], rcond=None)[0].reshape(-1)
    sol = uavoc.integrateSys(st["state"][0], horizon, theta)
    assert rel(sol["state_traj"], st["state"]) < 1e-10 and abs(sol["cost"] - float(st["cost"])) < 1e-9 * float(st["cost"])


def test_neural_policy_script_flow_cartpole(golden_dir):
    """Examples/OC/cartpole/cartpole_PDP_neural.py: init_step_neural_policy([n, n]); stored final parameters (column-major weights)"""
    from pdp_amd import PDP, JinEnv
    cartpole = JinEnv.CartPole()
    cartpole.initDyn(mc=0.1, mp=0.1, l=1)
    cartpole.initCost(wx=0.1, wq=0.6, wdx=0.1, wdq=0.1, wu=0.3)
    g = load(golden_dir, "oc_cartpole_neural.npz")
    oc = PDP.ControlPlanning()
    oc.setStateVariable(cartpole.X)
    oc.setControlVariable(cartpole.U)
    oc.setDyn(cartpole.X + float(g["dt"]) * cartpole.f)
    oc.setPathCost(cartpole.path_cost)
    oc.setFinalCost(cartpole.final_cost)
    oc.init_step_neural_policy(hidden_layers=[oc.n_state, oc.n_state])
    assert oc.n_auxvar == 45
    T = g["control"].shape[0]
    sol = oc.integrateSys(g["state"][0], T, g["param_final"])
    assert rel(sol["state_traj"], g["state"]) < 1e-10 and rel(sol["control_traj"], g["control"]) < 1e-10
    r = load(golden_dir, "ref_cp_cartpole_mlp.npz")
    loss, dp = oc.step(r["x0"], int(r["T"]), r["theta"])
    assert abs(loss - float(r["loss"])) < 1e-11 * abs(float(r["loss"])) and rel(dp, r["grad"]) < TOL
    u0 = oc.policy_fn(0, r["x0"], r["theta"]).full().flatten()              # host-side Function evaluation (casadi call semantics)
    assert rel(u0, r["control"][0]) < 1e-12


def test_irl_script_flow_cartpole_with_stored_demos(golden_dir):
    """Examples/IRL/cartpole/cartpole_PDP.py:9-29, 48-74 with the reference's stored demos as the learner's trajectory
    (theta = true parameter): getAuxSys -> LQR.set* -> lqrSolver -> chain rule, all through the drop-in classes."""
    from pdp_amd import PDP, JinEnv
    from pdp_amd.sx import vertcat
    cartpole = JinEnv.CartPole()
    cartpole.initDyn()
    cartpole.initCost(wu=0.1)
    d = load(golden_dir, "demos_cartpole.npz")
    dt = float(d["dt"])
    cartpoleoc = PDP.OCSys()
    cartpoleoc.setAuxvarVariable(vertcat(cartpole.dyn_auxvar, cartpole.cost_auxvar))
    cartpoleoc.setControlVariable(cartpole.U)
    cartpoleoc.setStateVariable(cartpole.X)
    dyn = cartpole.X + dt * cartpole.f
    cartpoleoc.setDyn(dyn)
    cartpoleoc.setPathCost(cartpole.path_cost)
    cartpoleoc.setFinalCost(cartpole.final_cost)
    cartpoleoc.diffPMP()
    lqr_solver = PDP.LQR()
    ra = load(golden_dir, "ref_auxsys_cartpole.npz")
    rl = load(golden_dir, "ref_lqr_cartpole.npz")
    i = 2
    aux_sys = cartpoleoc.getAuxSys(state_traj_opt=d["state"][i], control_traj_opt=d["control"][i], costate_traj_opt=d["costate"][i],
                                   auxvar_value=d["true_parameter"])
    assert set(aux_sys) == {"dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue", "hxx", "hxe"}
    for k in aux_sys:
        assert rel(np.stack(aux_sys[k]), ra[k][i]) < 1e-11, k
    lqr_solver.setDyn(dynF=aux_sys["dynF"], dynG=aux_sys["dynG"], dynE=aux_sys["dynE"])
    lqr_solver.setPathCost(Hxx=aux_sys["Hxx"], Huu=aux_sys["Huu"], Hxu=aux_sys["Hxu"], Hux=aux_sys["Hux"], Hxe=aux_sys["Hxe"], Hue=aux_sys["Hue"])
    lqr_solver.setFinalCost(hxx=aux_sys["hxx"], hxe=aux_sys["hxe"])
    aux_sol = lqr_solver.lqrSolver(np.zeros((cartpoleoc.n_state, cartpoleoc.n_auxvar)), d["control"].shape[1])
    assert set(aux_sol) == {"state_traj_opt", "control_traj_opt", "costate_traj_opt", "time"}
    assert rel(np.stack(aux_sol["state_traj_opt"]), rl["X"][i]) < TOL and rel(np.stack(aux_sol["control_traj_opt"]), rl["U"][i]) < TOL
    assert rel(np.stack(aux_sol["costate_traj_opt"]), rl["Lam"][i]) < TOL
    # host-side Function objects of diffPMP behave like casadi Functions
    F0 = cartpoleoc.dfx_fn(d["state"][i, 0], d["control"][i, 0], d["true_parameter"]).full()
    assert rel(F0, ra["dynF"][i, 0]) < 1e-12
    # ocSolver (one trajectory, the reference's dict) and ocSolver_batch (all demos at once) reach the stored IPOPT optima
    one = cartpoleoc.ocSolver(ini_state=d["state"][i, 0], horizon=d["control"].shape[1], auxvar_value=d["true_parameter"])
    assert set(one) == {"state_traj_opt", "control_traj_opt", "costate_traj_opt", "auxvar_value", "time", "horizon", "cost"}
    assert rel(one["state_traj_opt"], d["state"][i]) < 1e-6 and rel(one["costate_traj_opt"], d["costate"][i]) < 1e-6
    allb = cartpoleoc.ocSolver_batch(d["state"][:, 0], d["control"].shape[1], d["true_parameter"])
    assert bool(allb["converged"].all()) and rel(allb["state"].cpu().numpy(), d["state"]) < 1e-6
    assert rel(allb["cost"].cpu().numpy(), d["cost"]) < 1e-9


def test_lqr_class_input_polymorphism(golden_dir):
    """time-invariant ndarray inputs, omitted optional matrices, vector Hxe/Hue with 1-D ini_state (as ControlTools.iLQR.step uses it),
    wrong horizon -> AssertionError, singular Huu -> LinAlgError (reference PDP.py:339-555, 566)."""
    from pdp_amd import PDP
    r = load(golden_dir, "ref_lqr_random.npz")
    c = 4                                            # the time-invariant case of the reference run
    g = lambda k: r["c%d_%s" % (c, k)]
    lqr = PDP.LQR()
    lqr.setDyn(dynF=g("F")[0], dynG=g("G")[0], dynE=g("E")[0])
    lqr.setPathCost(Hxx=g("Hxx")[0], Huu=g("Huu")[0], Hxu=g("Hxu")[0], Hxe=g("Hxe")[0], Hue=g("Hue")[0])
    lqr.setFinalCost(hxx=[g("hxx")], hxe=[g("hxe")])
    sol = lqr.lqrSolver(g("X0"), int(g("T")))
    assert rel(np.stack(sol["state_traj_opt"]), g("X")) < TOL and rel(np.stack(sol["costate_traj_opt"]), g("Lam")) < TOL
    n, m = 3, 2
    lqr = PDP.LQR()
    lqr.setDyn(dynF=np.eye(n), dynG=np.ones((n, m)))
    lqr.setPathCost(Hxx=np.eye(n), Huu=np.eye(m), Hxe=np.ones((n, 1)), Hue=np.zeros((m, 1)))
    lqr.setFinalCost(hxx=np.eye(n), hxe=np.zeros((n, 1)))
    sol = lqr.lqrSolver([0.1, 0.2, 0.3], 4)
    assert sol["state_traj_opt"][0].shape == (n, 1) and len(sol["control_traj_opt"]) == 4
    lqr.setDyn(dynF=[np.eye(n)] * 3, dynG=np.ones((n, m)))
    with pytest.raises(AssertionError):
        lqr.lqrSolver([0.1, 0.2, 0.3], 4)
    lqr.setDyn(dynF=np.eye(n), dynG=np.zeros((n, m)))
    lqr.setPathCost(Hxx=np.eye(n), Huu=np.zeros((m, m)))
    with pytest.raises(np.linalg.LinAlgError):
        lqr.lqrSolver([0.1, 0.2, 0.3], 4)


def test_user_defined_model_is_generated_and_compiled_on_the_fly():
    """a system that is not in the zoo (Van der Pol oscillator with a learnable damping): symbolic -> HIP -> result vs finite differences"""
    from pdp_amd import PDP
    from pdp_amd.sx import SX, vertcat
    x1, x2, u, mu = SX.sym("x1"), SX.sym("x2"), SX.sym("u"), SX.sym("mu")
    X = vertcat(x1, x2)
    f = vertcat(x2, mu * (1 - x1 * x1) * x2 - x1 + u)
    sid = PDP.SysID("vanderpol test")
    sid.setAuxvarVariable(mu)
    sid.setStateVariable(X)
    sid.setControlVariable(u)
    sid.setDyn(X + 0.05 * f)
    rng = np.random.default_rng(0)
    inputs = [rng.uniform(-1, 1, (20, 1)) for _ in range(4)]
    states = [sid.integrateDyn([1.0, 0.0], inp, [0.8]) for inp in inputs]
    loss, d = sid.step(inputs, states, [0.5])
    eps = 1e-6
    lp, _ = sid.step(inputs, states, [0.5 + eps])
    lm, _ = sid.step(inputs, states, [0.5 - eps])
    assert loss > 0 and abs(2 * d[0] - (lp - lm) / (2 * eps)) < 1e-6 * abs(2 * d[0])       # step returns half the gradient (PDP.py:1285-1290)


def test_dense_numeric_linear_model_with_more_than_64_constants():
    """A user model whose Jacobians are all NUMERIC constants (x+ = x + dt (A x + B u), A 10x10 dense): the 120 distinct constants of
    F = I + dt A and G = dt B live in the kernels' LDS constant pools, which used to be filled by one lane-indexed pass (64 entries).
    OCSys: aux matrices and the fused gradient against numpy; ControlPlanning: gradient against finite differences."""
    from oracle import pdp_oracle as po
    from pdp_amd import PDP
    from pdp_amd.sx import SX, vertcat, mtimes, dot
    rng = np.random.default_rng(11)
    n, m, dt, T, B = 10, 2, 0.05, 12, 5
    A = rng.standard_normal((n, n)) - 1.5 * np.eye(n)
    Bm = rng.standard_normal((n, m))
    X, U, w = SX.sym("x", n), SX.sym("u", m), SX.sym("w", 2)
    f = X + dt * (mtimes(SX(A), X) + mtimes(SX(Bm), U))
    oc = PDP.OCSys("dense linear")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(X)
    oc.setControlVariable(U)
    oc.setDyn(f)
    oc.setPathCost(w[0] * dot(X, X) + w[1] * dot(U, U))
    oc.setFinalCost(w[0] * dot(X, X))
    th = np.array([0.7, 0.3])
    x0 = rng.standard_normal((B, n))
    u = 0.5 * rng.standard_normal((B, T, m))
    xs, _ = oc.rollout_batch(x0, u, th)
    lam = oc.costate_batch(xs, u, th)
    aux = oc.getAuxSys_batch(xs, u, lam, th)
    F, G = np.eye(n) + dt * A, dt * Bm
    xn = npy(xs)
    assert np.abs(npy(aux["dynF"]) - F).max() <= 1e-15 and np.abs(npy(aux["dynG"]) - G).max() <= 1e-15 and float(aux["dynE"].abs().max()) == 0.0
    assert np.abs(xn[:, 1] - (x0 @ F.T + u[:, 0] @ G.T)).max() <= 1e-14
    demo_x, demo_u = xn + 0.1 * rng.standard_normal(xn.shape), u + 0.1 * rng.standard_normal(u.shape)
    out = oc.pdp_grad_batch(u, th, demo_x, demo_u, ini_state=x0, want_sens=True)
    assert int(out["status"].sum()) == 0
    for i in range(B):
        a = {k: [np.asarray(v) for v in npy(aux[k])[i]] for k in ("dynF", "dynG", "dynE", "Hxx", "Hxu", "Hxe", "Hux", "Huu", "Hue")}
        a["hxx"], a["hxe"] = [npy(aux["hxx"])[i]], [npy(aux["hxe"])[i]]
        assert np.abs(a["Hxx"][3] - 2 * th[0] * np.eye(n)).max() <= 1e-15 and np.abs(a["Hue"][3] - np.stack([0 * u[i, 3], 2 * u[i, 3]], axis=1)).max() <= 1e-15
        ref = po.lqr_from_aux(a, n, 2, T)
        assert rel(npy(out["dxdp"])[i], np.stack(ref["state_traj_opt"])) < 1e-10
        l, g = po.irl_loss_grad(xn[i], u[i], demo_x[i], demo_u[i], ref["state_traj_opt"], ref["control_traj_opt"])
        assert abs(npy(out["loss"])[i] - l) <= 1e-12 * l and rel(npy(out["grad"])[i], g) < 1e-10
    cp = PDP.ControlPlanning("dense linear cp")
    cp.setStateVariable(X)
    cp.setControlVariable(U)
    cp.setDyn(f)
    cp.setPathCost(0.7 * dot(X, X) + 0.3 * dot(U, U))
    cp.setFinalCost(0.7 * dot(X, X))
    cp.init_step(T, n_poly=3)
    thp = 0.3 * rng.standard_normal(cp.n_auxvar)
    loss, grad = cp.step(x0[0], T, thp)
    eps = 1e-6
    for k in (0, cp.n_auxvar - 1):
        tp, tm = thp.copy(), thp.copy()
        tp[k] += eps
        tm[k] -= eps
        fd = (cp.step(x0[0], T, tp)[0] - cp.step(x0[0], T, tm)[0]) / (2 * eps)
        assert abs(fd - grad[k]) <= 1e-6 * max(1.0, abs(fd))


@pytest.mark.parametrize("fixture", ["ref_warp_pendulum_0", "ref_recmat_pendulum_0", "ref_warp_cartpole_1", "ref_recmat_rocket_2", "ref_recmat_quadrotor_3",
                                     "ref_recmat_long_rocket", "ref_recmat_long_quadrotor", "ref_recmat_long_robotarm"])
def test_warp_and_recmat_variants_match_reference_run(golden_dir, fixture):
    """ControlPlanning.warp_step / recmat_step / *_unwarp (PDP.py:882-1141): the reference composes the dynamics symbolically over
    grid cells; here the same gradient comes from one adjoint (costate) sweep on the GPU.  The rocket / quadrotor fixtures are the
    reference's own recmat_init_step(horizon, -1) runs (Examples/OC/rocket/rocket_PDP_Recmat.py:47-64, uav_PDP_Recmat.py: one cell per
    time step, their initial states), at the horizon its symbolic recovery matrix can still be composed by the build container's sympy stand-in
    (T = 7); the ref_recmat_long_* fixtures are the same reference code at the drivers' REAL horizons - rocket T = 50, quadrotor T = 35,
    robot arm T = 20 (robotarm_PDP_Recmat.py:49-56) - executed on this package's SX layer as the CasADi stand-in, cross-checked against the
    sympy runs (tests/golden/make_recmat_long.py)."""
    from pdp_amd import PDP, zoo
    g = load(golden_dir, fixture + ".npz")
    mode, name = fixture.split("_")[1], fixture.split("_")[-1] if "long" in fixture else fixture.split("_")[2]
    env, _ = zoo.make_env(name, "oc")
    cp = PDP.ControlPlanning()
    cp.setStateVariable(env.X)
    cp.setControlVariable(env.U)
    cp.setDyn(env.X + float(g["dt"]) * env.f)
    cp.setPathCost(env.path_cost)
    cp.setFinalCost(env.final_cost)
    T = int(g["T"])
    if mode == "warp":
        cp.warp_init_step(T)
        loss, grad = cp.warp_step(g["x0"], T, g["theta"])
        un = cp.warp_unwarp(g["x0"], T, g["theta"])
    else:
        cp.recmat_init_step(T) if int(g["grid"]) == -2 else cp.recmat_init_step(T, int(g["grid"]))
        loss, grad = cp.recmat_step(g["x0"], T, g["theta"])
        un = cp.recmat_unwarp(g["x0"], T, g["theta"])
    assert np.array_equal(cp.time_grid, g["time_grid"]) and cp.n_auxvar == g["theta"].size
    assert abs(loss - float(g["loss"])) <= 1e-11 * abs(float(g["loss"]))
    assert rel(grad, g["grad"]) < TOL
    assert rel(un["state_traj"], g["state"]) < 1e-11 and rel(un["control_traj"], g["control"]) < 1e-12
