"""CPU: the CasADi front-end adapter (pdp_amd/casadi_adapter.py) - a casadi.Function's SX instruction tape replayed on this package's expression DAG.
CasADi is absent from the build image: the walk is tested against RECORDED tapes (the same introspection calls answered from JSON, tests/golden/casadi_tape_*.json,
written by tests/golden/make_casadi_tape.py) and, where `import casadi` works, against the live module."""
import json
import os

import numpy as np
import pytest


def tape(golden_dir, name):
    from pdp_amd import casadi_adapter as ca
    return ca.RecordedTape(os.path.join(golden_dir, "casadi_tape_%s.json" % name))


def test_recorded_cartpole_tape_becomes_the_same_model(golden_dir):
    """the cart-pole IRL model arriving through a Function tape: same values, same derivatives - and, node for node, the same DAG as the native model, so the
    code generator produces the SAME device code (same content hash: the library built for the zoo model serves it)"""
    from pdp_amd import casadi_adapter as ca, codegen, sx, zoo
    t = tape(golden_dir, "cartpole")
    assert t.n_in() == 3 and t.n_out() == 3 and t.n_instructions() > 60
    env, dt = zoo.make_env("cartpole", "irl")
    th = sx.vertcat(env.dyn_auxvar, env.cost_auxvar)
    g = ca.from_casadi(t, inputs=[env.X, env.U, th])
    dyn, pc, fc = g.outs
    native = sx.Function("n", [env.X, env.U, th], [env.X + dt * env.f, env.path_cost, env.final_cost, sx.jacobian(env.X + dt * env.f, th)])
    conv = sx.Function("c", [env.X, env.U, th], [dyn, pc, fc, sx.jacobian(dyn, th)])
    rng = np.random.default_rng(0)
    for _ in range(5):
        x, u, e = rng.standard_normal(4), rng.standard_normal(1), np.array([0.5, 0.5, 1, 1, 6, 1, 1.0]) * (1 + 0.2 * rng.uniform(-1, 1, 7))
        for a, b in zip(native(x, u, e), conv(x, u, e)):
            assert np.array_equal(a.full(), b.full())
    pa = codegen.Problem(codegen.KIND_OC, env.X, env.U, dyn, th, pc, fc, label="cartpole")
    assert codegen.generate(pa)[1]["name"] == codegen.generate(zoo.make_problem("cartpole", "irl"))[1]["name"]
    # fresh symbols named after the Function's inputs when none are given
    h = ca.from_casadi(t)
    assert [i.numel() for i in h.ins] == [4, 1, 7] and h.ins[0].data[0].name.startswith("state")
    x, u, e = rng.standard_normal(4), rng.standard_normal(1), np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
    assert all(np.array_equal(a.full(), b.full()) for a, b in zip(h(x, u, e), native(x, u, e)[:3]))


def test_handwritten_tape_in_casadi_style(golden_dir):
    """work slots re-used, OP_SQ / OP_TWICE / OP_INV / OP_CONSTPOW, a sparse (compressed-column) output: closed-form known answers"""
    from pdp_amd import casadi_adapter as ca
    f = ca.from_casadi(tape(golden_dir, "handwritten"))
    x, p = np.array([0.7, -1.3]), np.array([2.5])
    o0, o1, o2 = (a.full() for a in f(x, p))
    assert np.allclose(o0[:, 0], [x[0] ** 2 + 2 * x[1] * p[0], np.sin(x[0]) / p[0]], rtol=1e-15, atol=0)
    assert np.allclose(o1, [[2 * x[0], 2 * p[0]], [np.cos(x[0]) / p[0], 0.0]], rtol=1e-15, atol=0) and o1[1, 1] == 0.0
    assert np.isclose(o2[0, 0], x[1] ** 3, rtol=1e-15)
    from pdp_amd import sx
    assert f.outs[1].data[3] is sx.ZERO                       # the structural zero of the sparse output is an exact zero of the DAG (code generation drops it)
    J = sx.Function("J", f.ins, [sx.jacobian(f.outs[0], f.ins[0])])(x, p).full()      # differentiating the converted DAG reproduces the tape's own Jacobian output
    assert np.allclose(J, o1, rtol=1e-15, atol=0)


def test_unknown_operations_and_bad_tapes_are_refused(golden_dir):
    from pdp_amd import casadi_adapter as ca
    d = json.load(open(os.path.join(golden_dir, "casadi_tape_handwritten.json")))
    d["instructions"][3][0] = 33
    d["op_names"]["33"] = "OP_ERF"
    with pytest.raises(NotImplementedError, match="OP_ERF"):
        ca.from_casadi(ca.RecordedTape(d))
    d = json.load(open(os.path.join(golden_dir, "casadi_tape_handwritten.json")))
    d["instructions"][4][1] = [1, 4]                           # reads a work slot nothing has written yet
    with pytest.raises(AssertionError, match="unset work slot"):
        ca.from_casadi(ca.RecordedTape(d))
    with pytest.raises(AssertionError, match="inputs"):
        ca.from_casadi(tape(golden_dir, "handwritten"), inputs=[])


def test_tape_export_round_trip_of_every_zoo_oc_model():
    """tape_of -> JSON -> RecordedTape -> from_casadi is the identity on the DAG for the dynamics and costs of all five systems (pow, tan, sqrt ... included)"""
    from pdp_amd import casadi_adapter as ca, sx, zoo
    for system in ("pendulum", "cartpole", "robotarm", "quadrotor", "rocket"):
        env, dt = zoo.make_env(system, "irl")
        th = sx.vertcat(env.dyn_auxvar, env.cost_auxvar)
        f = sx.Function(system, [env.X, env.U, th], [env.X + dt * env.f, env.path_cost, env.final_cost])
        g = ca.from_casadi(ca.RecordedTape(json.loads(json.dumps(ca.tape_of(f)))), inputs=[env.X, env.U, th])
        for a, b in zip(f.outs, g.outs):
            assert all(p is q for p, q in zip(a.data, b.data)), system      # hash-consed: the very same nodes


def test_live_casadi_function():
    """with CasADi installed: the real module's tape of a small OC model, recorded and converted, evaluates like the Function itself; OCSys accepts casadi.SX"""
    casadi = pytest.importorskip("casadi")
    from pdp_amd import casadi_adapter as ca
    x, u, p = casadi.SX.sym("x", 2), casadi.SX.sym("u", 1), casadi.SX.sym("p", 3)
    dyn = x + 0.1 * casadi.vertcat(x[1], (u[0] - p[0] * casadi.sin(x[0]) - p[1] * x[1]) / p[2])
    cost = casadi.dot(x, x) + 0.1 * casadi.dot(u, u)
    fn = casadi.Function("pend", [x, u, p], [dyn, cost, casadi.jacobian(dyn, x)])
    g = ca.from_casadi(fn)
    h = ca.from_casadi(ca.RecordedTape(json.loads(json.dumps(ca.record(fn)))))
    rng = np.random.default_rng(1)
    xv, uv, pv = rng.standard_normal(2), rng.standard_normal(1), np.array([1.0, 0.1, 2.0])
    for a, b, c in zip(fn(xv, uv, pv), g(xv, uv, pv), h(xv, uv, pv)):
        assert np.allclose(np.asarray(a.full()), b.full(), rtol=1e-14, atol=1e-300) and np.array_equal(b.full(), c.full())
    from pdp_amd import PDP
    oc = PDP.OCSys("casadi_pendulum")
    oc.setAuxvarVariable(p)
    oc.setStateVariable(x)
    oc.setControlVariable(u)
    oc.setDyn(dyn)
    oc.setPathCost(cost)
    oc.setFinalCost(casadi.dot(x, x))
    assert np.allclose(oc.dyn_fn(xv, uv, pv).full().ravel(), np.asarray(fn(xv, uv, pv)[0].full()).ravel(), rtol=1e-14)


# ---- round 6: CasADi objects at the ControlPlanning and SysID class surface (PDP/PDP.py:672-697, 1178-1188 take casadi.SX) ------------------------------------------
class _ReplaySX:
    """What the class surface sees of a casadi.SX: a type that lives in module `casadi`, numel(), and - for an expression - the recorded tape of the casadi.Function
    the reference would wrap it in.  CasADi is absent from the build image: the stand-in answers `casadi.Function(name, vars, [expr])` with that tape, so that
    _CasadiFrontEnd._own / _own_expr, the variable bookkeeping and casadi_adapter.convert_expression run exactly as with the real module."""
    __module__ = "casadi"

    def __init__(self, n, tape=None, deps=()):
        self.n, self.tape, self.deps = n, tape, tuple(deps)

    def numel(self):
        return self.n


def _replay_casadi(monkeypatch):
    import sys
    import types
    from pdp_amd import casadi_adapter as ca
    mod = types.ModuleType("casadi")

    def Function(name, ins, outs):
        (expr,) = outs
        assert isinstance(expr, _ReplaySX) and expr.tape is not None
        assert len(ins) == len(expr.deps) and all(a is b for a, b in zip(ins, expr.deps)), "the Function is built over the variables the expression was recorded in"
        return ca.RecordedTape(dict(expr.tape, name=name))
    mod.Function = Function
    mod.SX = _ReplaySX
    monkeypatch.setitem(sys.modules, "casadi", mod)
    return mod


def _same_model(info, native):
    """The class surface mirrors the CasADi symbols by FRESH symbols of its own, and sx.py orders the operands of commutative operations by node id: the generated
    source then lists `u[1] + u[0]` where the native model has `u[0] + u[1]` - another content hash, the same arithmetic bit for bit (the tests compare values and
    Jacobians with array_equal).  What must agree is everything the code generator derives from the DAG: sizes, distinct non-zero entries and constants of every
    matrix group, theta-only precomputed values, operation counts, LDS chunking.  (On the native symbols the hash itself agrees: the cart-pole test above.)"""
    assert {k: v for k, v in info.items() if k != "name"} == {k: v for k, v in native.items() if k != "name"}


def test_casadi_objects_at_the_controlplanning_surface(golden_dir, monkeypatch):
    """ControlPlanning built from casadi.SX symbols and expressions (recorded tapes of the quadrotor model of Examples/OC/quadrotor/uav_PDP.py:9-21): the model that
    arrives is the native one (_same_model; values and Jacobians bit for bit), policies and Jacobian Functions are built on the mirrored symbols."""
    from pdp_amd import PDP, codegen, sx, zoo
    _replay_casadi(monkeypatch)
    tapes = json.load(open(os.path.join(golden_dir, "casadi_tape_quadrotor_cp.json")))
    X, U = _ReplaySX(13), _ReplaySX(4)
    cp = PDP.ControlPlanning("quadrotor")
    cp.setStateVariable(X)
    cp.setControlVariable(U)
    assert isinstance(cp.state, sx.SX) and cp.n_state == 13 and cp.n_control == 4 and cp._casadi_vars == {"state": X, "control": U}
    cp.setDyn(_ReplaySX(13, tapes["dyn"], (X, U)))
    cp.setPathCost(_ReplaySX(1, tapes["path_cost"], (X, U)))
    cp.setFinalCost(_ReplaySX(1, tapes["final_cost"], (X,)))
    pb = codegen.Problem(codegen.KIND_CP, cp.state, cp.control, cp.dyn, None, cp.path_cost, cp.final_cost, label="quadrotor")
    _same_model(codegen.generate(pb)[1], codegen.generate(zoo.make_problem("quadrotor", "oc"))[1])
    env, dt = zoo.make_env("quadrotor", "oc")
    native = sx.Function("n", [env.X, env.U], [env.X + dt * env.f, env.path_cost, sx.jacobian(env.X + dt * env.f, env.X), sx.jacobian(env.path_cost, env.U)])
    rng = np.random.default_rng(3)
    x, u = rng.standard_normal(13), rng.standard_normal(4)
    for a, b in zip(native(x, u), (cp.dyn_fn(x, u), cp.path_cost_fn(x, u), cp.dfx_fn(x, u), cp.dcu_fn(x, u))):
        assert np.array_equal(a.full(), b.full())
    cp.init_step(50)                                               # the Lagrange policy is stated on the mirrored symbols
    assert cp.n_auxvar == 24
    # an own sx object after a CasADi one drops the stale mapping; a CasADi expression without CasADi variables is refused
    cp.setControlVariable(sx.SX.sym("u", 4))
    assert "control" not in cp._casadi_vars
    cp2 = PDP.ControlPlanning("x")
    cp2.setStateVariable(sx.SX.sym("x", 13))
    cp2.setControlVariable(sx.SX.sym("u", 4))
    with pytest.raises(AssertionError, match="no variable was given"):
        cp2.setDyn(_ReplaySX(13, tapes["dyn"], (X, U)))


def test_casadi_objects_at_the_sysid_surface(golden_dir, monkeypatch):
    """SysID.setDyn with casadi.SX (PDP.py:1178-1188): the quadrotor model of Examples/SysID/quadrotor through its recorded tape = the native model (_same_model; f, f_x,
    f_theta bit for bit)."""
    from pdp_amd import PDP, codegen, sx, zoo
    _replay_casadi(monkeypatch)
    tape = json.load(open(os.path.join(golden_dir, "casadi_tape_quadrotor_sysid.json")))["dyn"]
    X, U, P = _ReplaySX(13), _ReplaySX(4), _ReplaySX(5)
    sid = PDP.SysID("quadrotor")
    sid.setAuxvarVariable(P)
    sid.setStateVariable(X)
    sid.setControlVariable(U)
    sid.setDyn(_ReplaySX(13, tape, (X, U, P)))
    pb = codegen.Problem(codegen.KIND_SYSID, sid.state, sid.control, sid.dyn, sid.auxvar, label="quadrotor")
    _same_model(codegen.generate(pb)[1], codegen.generate(zoo.make_problem("quadrotor", "sysid"))[1])
    env, dt = zoo.make_env("quadrotor", "sysid")
    dyn = env.X + dt * env.f
    native = sx.Function("n", [env.X, env.U, env.dyn_auxvar], [dyn, sx.jacobian(dyn, env.X), sx.jacobian(dyn, env.dyn_auxvar)])
    rng = np.random.default_rng(4)
    x, u, th = rng.standard_normal(13), rng.standard_normal(4), np.array([1.0, 1.1, 0.9, 1.2, 0.4])
    for a, b in zip(native(x, u, th), (sid.dyn_fn(x, u, th), sid.dfx_fn(x, u, th), sid.dfe_fn(x, u, th))):
        assert np.array_equal(a.full(), b.full())
