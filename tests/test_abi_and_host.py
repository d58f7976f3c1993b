"""CPU (no GPU needed): the C-ABI libraries load and export every symbol include/pdp_hip.h declares; code generation is
deterministic; the symbolic engine's derivatives agree with sympy; argument validation happens before any launch."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "pdp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pdp_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from pdp_amd import codegen, zoo
    return codegen, zoo


def test_every_declared_symbol_is_exported(built):
    codegen, zoo = built
    from pdp_amd import runtime
    syms = _declared_symbols()
    assert set(runtime.CORE_SYMBOLS + runtime.MODEL_SYMBOLS + ["pdp_cp_grad_contract_batched"]) == set(syms)
    core = ctypes.CDLL(os.path.join(ROOT, "pontryagin-differentiable-programming_amd", "lib", "libpdp_hip.so"))
    core_syms = [s for s in syms if s in runtime.CORE_SYMBOLS or s == "pdp_cp_grad_contract_batched"]
    for s in core_syms:
        assert hasattr(core, s), s
    core.pdp_hip_version.restype = ctypes.c_char_p
    assert b"gfx950" in core.pdp_hip_version()
    lib_path, info = codegen.build_problem(zoo.make_problem("cartpole", "irl"))
    mdl = ctypes.CDLL(lib_path)
    for s in runtime.MODEL_SYMBOLS:
        assert hasattr(mdl, s), s


def test_model_info_and_argument_validation_without_gpu(built):
    codegen, zoo = built
    from pdp_amd import runtime
    m = runtime.load_model(codegen.build_problem(zoo.make_problem("quadrotor", "irl"))[0])
    assert (m.kind, m.n, m.m, m.p) == (0, 13, 4, 9) and m.name.startswith("quadrotor_oc_")
    # bad arguments are rejected by the C entry points before anything is launched (no GPU needed)
    assert m.lib.pdp_oc_rollout_batched(0, 10, None, None, None, 0, None, None, None) == -1
    assert m.lib.pdp_oc_pdp_grad_batched(4, 10, 0, None, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert m.lib.pdp_sysid_step_batched(4, 10, None, None, None, 0, None, None, None) == -4        # wrong model kind
    core = runtime.load_core()
    assert core.pdp_lqr_workspace_bytes(2, 5, 3, 1, 4, 1) == 2 * 5 * (3 * 1 + 1 * 4 + 3 * 3 + 3 * 4) * 8
    pr = runtime.PdpLqrProblem()
    pr.B, pr.T, pr.n, pr.m, pr.p = 1, 5, 20, 1, 1
    assert core.pdp_lqr_solve_batched(ctypes.byref(pr), None, None, None, None, None, 0, None) == -1


def test_workspace_size_entry_points_are_host_side_and_consistent(built):
    """the *_workspace_bytes entry points are plain host functions (callable without a GPU) and follow the documented layouts"""
    codegen, zoo = built
    from pdp_amd import runtime
    oc = runtime.load_model(codegen.build_problem(zoo.make_problem("quadrotor", "irl"))[0])
    B, T, n, m, p = 7, 50, 13, 4, 9
    assert oc.lib.pdp_oc_pdp_workspace_bytes(B, T) == B * T * (m * n + m * p + 1) * 8          # K, k and the zero sink slot per step
    small, big = oc.lib.pdp_oc_solve_workspace_bytes(B, T, 4), oc.lib.pdp_oc_solve_workspace_bytes(B, T, 10)
    assert 0 < small < big and big - small >= 6 * B * ((T + 1) * n + T * m) * 8              # trial trajectories grow with ls_trials
    assert oc.lib.pdp_oc_solve_workspace_bytes(2 * B, T, 10) > big
    assert oc.lib.pdp_cp_step_workspace_bytes(B, T, None, 0) == 0                              # not a ControlPlanning model
    cp = runtime.load_model(codegen.build_problem(zoo.make_problem("quadrotor", "oc"))[0])
    mlp = runtime.make_policy("mlp", layers=[13, 13, 4])
    poly = runtime.make_policy("poly", pivots=np.linspace(0, 100, 6))
    # MLP [13,13] runs on the register-resident kernels: the larger of one stored activation per lane and time step (one trajectory per wavefront) and the
    # four-trajectory kernel's activations in D layout + the trajectories (cp_mlp4t_ws_doubles)
    wsb = lambda b, t: 8 * max(b * t * 64, ((b + 3) // 4) * t * 3 * 64 + b * ((t + 1) * 13 + t * 4))
    assert cp.lib.pdp_cp_step_workspace_bytes(16, 100, ctypes.byref(mlp), 420) == wsb(16, 100)
    assert cp.lib.pdp_cp_step_workspace_bytes(1024, 100, ctypes.byref(mlp), 420) == wsb(1024, 100)
    # policies beyond every tuned kernel take the size-generic one: trajectory + controls + the hidden activations of every step, per trajectory
    huge = runtime.make_policy("mlp", layers=[64, 64, 4])
    assert cp.lib.pdp_cp_step_workspace_bytes(8, 100, ctypes.byref(huge), 64 * 13 + 64 + 64 * 64 + 64 + 4 * 64 + 4) == 8 * (101 * 13 + 100 * 4 + 100 * 128) * 8
    # a network beyond that kernel (width > 16): the general adjoint kernel, 20 stored activations per step, offloaded only when the batch does not
    # fit the CUs at once (256 CUs assumed without a GPU)
    wide = runtime.make_policy("mlp", layers=[20, 4])
    assert cp.lib.pdp_cp_step_workspace_bytes(16, 100, ctypes.byref(wide), 20 * 13 + 20 + 4 * 20 + 4) == 0
    assert cp.lib.pdp_cp_step_workspace_bytes(1024, 100, ctypes.byref(wide), 20 * 13 + 20 + 4 * 20 + 4) == 1024 * 100 * 20 * 8
    assert cp.lib.pdp_cp_step_workspace_bytes(1024, 100, ctypes.byref(poly), 24) == 0
    assert cp.lib.pdp_oc_solve_workspace_bytes(B, T, 10) == 0 and cp.lib.pdp_oc_pdp_workspace_bytes(B, T) == 0
    # argument validation of the solver entry point happens before any launch
    assert oc.lib.pdp_oc_solve_batched(0, T, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert cp.lib.pdp_oc_solve_batched(B, T, None, None, 0, None, None, None, None, None, None, None, None, None, None, 0, None) == -4


def test_codegen_is_deterministic_and_cached(built):
    codegen, zoo = built
    a = codegen.generate(zoo.make_problem("rocket", "irl"))
    b = codegen.generate(zoo.make_problem("rocket", "irl"))
    assert a[0] == b[0] and a[1]["name"] == b[1]["name"]
    assert os.path.exists(codegen.lib_path(a[1]["name"]))


def test_tracked_generated_headers_are_current_and_chunks_follow_the_lds_budget(built):
    """the generated headers kept in the tree are what the generator produces today (a stale header would ship old device code),
    and the chunk lengths follow the documented LDS rule: fused OC models take as many pool rows as fit 40 KB together with the
    kernel's other LDS, the other kinds a power of two within 34 KB"""
    codegen, zoo = built
    for spec in zoo.SPECS:
        pb = zoo.make_problem(*spec)
        src, info = codegen.generate(pb)
        assert open(codegen.header_path(info["name"])).read() == src, spec
        nv = info["nvar"]
        if info["kind"] == codegen.KIND_OC:
            stride = (nv["patha"] + nv["pathb"] + info["n"]) | 1
            assert 2 <= info["chunk"] <= 64 and info["chunk"] * stride * 8 <= 40 * 1024
            assert info["chunk"] == 64 or (info["chunk"] + 1) * stride * 8 + 8 * (608 + info["n"] + info["p"] + info["npc"]) > 40 * 1024
        else:
            assert info["chunk"] in (2, 4, 8, 16, 32) and info["chunk"] * (nv["path"] | 1) * 8 <= 34 * 1024
    assert codegen._pick_chunk(181 + 13, 0) == 16 and codegen._pick_chunk(86 + 107, 13, other_doubles=608 + 4 + 13 + 9 + 23 + 8) == 21


def test_tuned_name_list_is_static_and_current(built):
    """codegen.TUNED_NAMES is complete at import (read from the tracked csrc/generated/tuned_names.txt), so the flag set of a model does not depend on
    whether zoo.register_tuned() has run yet in the process (round-3 advisor finding), and the file is what the generator produces today"""
    import subprocess
    import sys
    codegen, zoo = built
    names = zoo.tuned_names()
    assert codegen._read_tuned() == set(names) and len(names) == len(zoo.SPECS)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from pdp_amd import codegen; print(codegen.tuned(%r), codegen.tuned('quadrotor_oc_0000000000'))"
                          % (root, names[0])], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert out == ["True", "False"]                 # a fresh process, zoo never imported


FAST_PATH = ("oc_pdp_fused3_kernel", "oc_pdp_fused_kernel", "oc_solve_ms2_kernel", "oc_solve_ms_kernel", "oc_auxsys_kernel", "oc_predict_kernel", "cp_step_poly_kernel",
             "cp_step_poly2_kernel", "cp_step_mlp16_kernel", "sysid_step_kernel", "sysid_step2_kernel", "lqr_solve_kernel", "lqr_solve_stream_kernel", "lqr_solve_small_kernel")


def test_no_kernel_spills_registers(built):
    """Read from the shipped code objects (codegen.kernel_resources: llvm-readelf --notes): NO kernel of any zoo library or of the core library spills
    vector registers, and the fast-path kernels use no scratch memory at all.  (Round 3 shipped oc_solve_ms2_kernel<.., 4> with 21 spilled VGPRs, the
    single-shooting helpers with 206 and lqr_solve_stream_kernel with 46 - 50.)  Scratch WITHOUT spills remains only where a kernel indexes a local array by
    a run-time policy description (the materialised ControlPlanning drop-ins and the general MLP kernel: layer sizes are kernel arguments)."""
    import glob
    codegen, zoo = built
    keep = set(zoo.tuned_names())
    libs = [codegen.CORE_LIB_PATH] + [p for p in sorted(glob.glob(os.path.join(codegen.LIB_DIR, "libpdp_model_*.so")))
                                      if os.path.basename(p)[len("libpdp_model_"):-3].replace("__plain", "") in keep]
    assert len(libs) >= 1 + len(keep)
    seen = set()
    for lib in libs:
        res = codegen.kernel_resources(lib)
        if any(k.startswith("lqr_solve_small_kernel") and v["scratch"] for k, v in res.items()):
            res = codegen.kernel_resources(lib, count_scratch_instructions=True)
        for name, r in res.items():
            base = name.split("<")[0]
            seen.add(base)
            assert r["spill"] == 0, (os.path.basename(lib), name, r)
            if base == "lqr_solve_small_kernel":
                # 20 bytes of DEAD stack in three instantiations (spill slots of scalar registers the allocator afterwards placed in vector-register lanes):
                # the metadata keeps the frame size, the code contains no scratch access - checked in the disassembly
                assert r["scratch"] <= 32 and r.get("scratch_instructions", 0) == 0, (name, r)
            elif base in FAST_PATH:
                assert r["scratch"] == 0, (os.path.basename(lib), name, r)
            else:
                assert r["scratch"] == 0 or base in ("cp_integrate_kernel", "cp_auxsys_kernel", "cp_step_adjoint_kernel"), (os.path.basename(lib), name, r)
    assert {"oc_pdp_fused3_kernel", "oc_solve_ms2_kernel", "lqr_solve_stream_kernel", "cp_step_mlp16_kernel", "sysid_step2_kernel"} <= seen


def test_ocsys_setters_drop_the_compiled_models():
    """changing the cost, the dynamics or the bounds after a solve must not reuse the old (barrier) model (round-3 advisor finding)"""
    from pdp_amd import PDP, sx
    oc = PDP.OCSys()
    x, u = sx.SX.sym("x", 2), sx.SX.sym("u", 1)
    oc.setStateVariable(x)
    oc.setControlVariable(u, [-1.0], [1.0])
    oc.setDyn(x + 0.1 * sx.vertcat(x[1], u[0]))
    oc.setPathCost(sx.dot(x, x) + sx.dot(u, u))
    oc.setFinalCost(sx.dot(x, x))
    for setter, args in ((oc.setPathCost, (2 * sx.dot(x, x) + sx.dot(u, u),)), (oc.setFinalCost, (3 * sx.dot(x, x),)), (oc.setDyn, (x + 0.2 * sx.vertcat(x[1], u[0]),)),
                         (oc.setControlVariable, (u, [-2.0], [2.0])), (oc.setStateVariable, (x, [-5.0, -5.0], [5.0, 5.0])), (oc.setAuxvarVariable, (None,))):
        oc._model, oc._bar_model = "compiled", "compiled barrier"
        setter(*args)
        assert oc._model is None and oc._bar_model is None, setter.__name__


@pytest.mark.parametrize("system", ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
def test_symbolic_engine_derivatives_match_sympy(system):
    """the product's SX engine (used for code generation) against the independent sympy models of the oracle"""
    from oracle import models, pdp_oracle as po
    from pdp_amd import sx, zoo
    env, dt = zoo.make_env(system, "irl")
    th = sx.vertcat(env.dyn_auxvar, env.cost_auxvar)
    dyn = env.X + dt * env.f
    lam = sx.SX.sym("lam", env.X.numel())
    H = env.path_cost + sx.dot(dyn, lam)
    dHx = sx.jacobian(H, env.X).T
    fns = sx.Function("f", [env.X, env.U, lam, th], [dyn, sx.jacobian(dyn, env.X), sx.jacobian(dyn, th), sx.jacobian(dHx, env.X), sx.jacobian(dHx, th),
                                                      sx.jacobian(sx.jacobian(H, env.U).T, th), sx.jacobian(sx.jacobian(env.final_cost, env.X).T, th)])
    st = models.IRL_SETUP[system]
    oc = po.make_oc(models.REGISTRY[system](**st["kwargs"]), st["dt"])
    rng = np.random.default_rng(4)
    x, u, l = rng.standard_normal(oc.n), rng.standard_normal(oc.m), rng.standard_normal(oc.n)
    e = np.abs(rng.standard_normal(oc.p)) + 0.5
    got = [g.full() for g in fns(x, u, l, e)]
    ref = [np.asarray(oc.dyn_fn(x, u, e)).reshape(-1, 1), oc._m(oc.dfx_fn(x, u, e), oc.n, oc.n), oc._m(oc.dfe_fn(x, u, e), oc.n, oc.p),
           oc._m(oc.ddHxx_fn(x, u, l, e), oc.n, oc.n), oc._m(oc.ddHxe_fn(x, u, l, e), oc.n, oc.p), oc._m(oc.ddHue_fn(x, u, l, e), oc.m, oc.p),
           oc._m(oc.ddhxe_fn(x, e), oc.n, oc.p)]
    for a, b in zip(got, ref):
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max())


def test_sx_casadi_semantics():
    from pdp_amd import sx
    A = sx.SX.sym("A", 2, 3)
    v = A.reshape((-1, 1))
    assert v.shape == (6, 1) and v[1].data[0] is A[1, 0].data[0]          # column-major vec (PDP.py:740 relies on it)
    x = sx.SX.sym("x", 3)
    y = np.array([1.0, 2.0, 3.0]) - x                                        # numpy on the left defers to SX
    assert isinstance(y, sx.SX) and y.shape == (3, 1)
    f = sx.Function("f", [x], [sx.mtimes(A.T, sx.SX.sym("z", 2)) if False else sx.dot(x, x) * x])
    out = f([1.0, 2.0, 2.0]).full()
    assert np.allclose(out.flatten(), 9.0 * np.array([1.0, 2.0, 2.0]))
    J = sx.jacobian(sx.tanh(x[0] * x[1]) + x[2] ** 2, x)
    Jf = sx.Function("J", [x], [J])
    assert np.allclose(Jf([0.3, -0.2, 1.5]).full(), [[-0.2 * (1 - np.tanh(-0.06) ** 2), 0.3 * (1 - np.tanh(-0.06) ** 2), 3.0]])
    assert float(sx.inv(sx.SX(np.array([[2.0, 0.0], [0.0, 4.0]])))[1, 1]) == 0.25


def test_ocsolver_recognises_finite_bounds_and_new_entry_points_validate_arguments(built):
    """(i) the reference passes state / control bounds to IPOPT as lbw / ubw (PDP.py:141-168): finite bounds are recognised (the +-1e20 defaults mean
    "none", as for IPOPT) and routed to the barrier continuation (tests/test_gpu_ocsolver.py solves a bounded problem on the GPU);
    (ii) pdp_oc_solve_ms_batched / its workspace entry point are host-side and validate their arguments without a GPU."""
    from pdp_amd import PDP, runtime, zoo
    from pdp_amd.sx import SX, vertcat
    x, u, w = SX.sym("x", 2), SX.sym("u"), SX.sym("w")
    oc = PDP.OCSys("bounded")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(x)
    oc.setControlVariable(u, control_lb=[-1.0], control_ub=[1.0])
    oc.setDyn(x + 0.1 * vertcat(x[1], u))
    oc.setPathCost(w * (x[0] * x[0] + u * u))
    oc.setFinalCost(x[0] * x[0])
    assert oc.has_bounds()
    oc.setControlVariable(u)                      # defaults: +-1e20
    assert not oc.has_bounds()
    lib, info = built[0].build_problem(zoo.make_problem("pendulum", "irl"))
    m = runtime.ModelLib(lib)
    B, T = 7, 30
    n0, n1 = m.lib.pdp_oc_solve_ms_workspace_bytes(B, T, 0), m.lib.pdp_oc_solve_ms_workspace_bytes(B, T, 100)
    # either kernel variant may serve a call: the larger of the two layouts.  One-wave kernel: dx du dlam c gradx gradu gains P,W; runner / evaluator
    # kernel: nine stage-minor groups of (2 n + m) (T + 1) doubles (two point sets, the step, two residual sets, c_soc, the step kept during corrections, the watchdog's
    # stored iterate and direction) + gains (K | k) + (P | W)
    per1 = (T + 1) * 2 + T * 1 + T * 2 + T * 2 + (T + 1) * 2 + T * 1 + T * (2 * 1 + 1 + 1) + T * (4 + 2 + 1)
    per2 = 9 * (2 * 2 + 1) * (T + 1) + T * (2 * 1 + 1) + T * (4 + 2)
    assert n0 == 8 * B * (max(per1, per2) + 2) and n1 - n0 == 8 * B * 2 * 100
    opts = runtime.PdpOcMsOpts(1e-10, 100, 0, 0)
    import ctypes
    assert m.lib.pdp_oc_solve_ms_batched(0, T, None, None, 0, None, None, None, None, None, None, None, None, None, None, ctypes.byref(opts), None, 0, None) == -1
    cp = runtime.ModelLib(built[0].build_problem(zoo.make_problem("pendulum", "oc"))[0])
    assert cp.lib.pdp_oc_solve_ms_workspace_bytes(B, T, 10) == 0


def test_warp_and_recovery_matrix_internals_on_the_class_surface(golden_dir):
    """ControlPlanning.warp_dynCost / warp_getAuxSys / recmat_recoveryMatrix (PDP.py:882-958, 1039-1079) exist for callers that use the reference's
    symbolic internals directly (host-side objects on this package's SX layer; warp_step / recmat_step themselves run on the GPU).  The recovery matrix
    evaluated at the reference's own run (ref_recmat_pendulum_0.npz, sympy stand-in) must give the reference's gradient; the per-cell Jacobians must be
    the derivatives of the per-cell maps."""
    import numpy as np
    from pdp_amd import PDP, zoo
    g = np.load(os.path.join(golden_dir, "ref_recmat_pendulum_0.npz"))
    env, _ = zoo.make_env("pendulum", "oc")
    cp = PDP.ControlPlanning()
    cp.setStateVariable(env.X)
    cp.setControlVariable(env.U)
    cp.setDyn(env.X + float(g["dt"]) * env.f)
    cp.setPathCost(env.path_cost)
    cp.setFinalCost(env.final_cost)
    tg = g["time_grid"]
    cp.warp_dynCost(tg)
    W = len(tg) - 1
    assert len(cp.wdyn_fns) == W == len(cp.wdfx_fns) == len(cp.wdcu_fns)
    cp.recmat_recoveryMatrix(W)
    assert cp.n_auxvar == g["theta"].size
    grad = cp.recovery_matrix_fn(g["x0"], g["theta"]).full().flatten()
    assert np.abs(grad - g["grad"]).max() <= 1e-11 * np.abs(g["grad"]).max()
    # cell maps: cost and end state of the composed cells reproduce the stored rollout; Jacobians against central differences
    x, cost = g["x0"].copy(), 0.0
    for wt in range(W):
        u = g["theta"][wt:wt + 1]
        cost += float(cp.wpath_cost_fns[wt](x, u))
        x = cp.wdyn_fns[wt](x, u).full().flatten()
        assert np.abs(x - g["state"][tg[wt + 1]]).max() <= 1e-12
    assert abs(cost + float(cp.wfinal_cost_fn(x)) - float(g["loss"])) <= 1e-11 * abs(float(g["loss"]))
    x, u, eps = g["state"][tg[2]], g["theta"][2:3], 1e-6
    F = cp.wdfx_fns[2](x, u).full()
    for k in range(2):
        e = np.zeros(2)
        e[k] = eps
        fd = (cp.wdyn_fns[2](x + e, u).full().flatten() - cp.wdyn_fns[2](x - e, u).full().flatten()) / (2 * eps)
        assert np.abs(F[:, k] - fd).max() <= 1e-7
    cp.warp_init_step(int(g["T"]))
    cp.warp_dynCost(cp.time_grid)
    ws = np.stack([g["state"][t] for t in cp.time_grid])
    aux = cp.warp_getAuxSys(ws, np.zeros((cp.whorizon, 1)), np.zeros(cp.n_auxvar))
    assert len(aux["wdynF"]) == cp.whorizon and aux["wdynF"][0].shape == (2, 2) and aux["wdUe"][0].shape == (1, cp.n_auxvar) and np.all(aux["wdUx"][0] == 0)


def test_product_path_fails_loudly_without_a_gpu(built):
    """no CPU fallback anywhere on the product path: without a visible GPU the device-resident IRL loop (and every ModelLib call under it) raises instead of computing
    something somewhere else"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pdp_amd import zoo
    from pdp_amd.irl import IRLLoop
    mdl = zoo.get("cartpole", "irl")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        IRLLoop(mdl, np.zeros((2, 6, 4)), np.zeros((2, 5, 1)), np.ones(7), 1e-4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mdl.oc_solve_ms(np.zeros((2, 4)), np.ones(7), 5)


def test_library_stamps_cover_every_included_header(tmp_path):
    """Round-5 advice: pdp_model.hip gained `#include "pdp_cp_generic_kernels.h"` while the hand-written dependency list behind the model libraries' content-hash
    stamp did not - edits to that header left stale libraries.  The list is now the include closure of the translation unit: every local header either .hip file
    names (directly or through another header) is in it, and touching any of them changes the stamp."""
    from pdp_amd import codegen
    for unit in ("pdp_model.hip", "pdp_lqr.hip"):
        src = os.path.abspath(os.path.join(codegen.CSRC, unit))
        deps = codegen.source_closure(src)
        by_name = {os.path.basename(d): d for d in deps}
        assert deps[0] == src and "pdp_hip.h" in by_name
        for f in deps:                                           # every quoted include of every file of the list is itself in the list
            for inc in re.findall(r'#\s*include\s+"([^"]+)"', open(f).read()):
                assert os.path.basename(inc) in by_name, "%s includes %s, which the stamp of its library does not cover" % (os.path.basename(f), inc)
    assert "pdp_cp_generic_kernels.h" in {os.path.basename(d) for d in codegen.source_closure(os.path.join(codegen.CSRC, "pdp_model.hip"))}
    # the stamp is a content hash over that list: one byte more in any header is another stamp
    a, b = tmp_path / "a.h", tmp_path / "b.h"
    a.write_text('#include "b.h"\nint a;\n')
    b.write_text("int b;\n")
    d0 = codegen.source_closure(str(a))
    assert [os.path.basename(x) for x in d0] == ["a.h", "b.h"]
    s0 = codegen._stamp_of(d0, ["-O3"])
    b.write_text("int b; \n")
    assert codegen._stamp_of(codegen.source_closure(str(a)), ["-O3"]) != s0


def test_no_result_changing_macros_in_product_headers():
    """Round-5 verdict, item 8: build-time switches that change what a kernel computes (or make it fault) live in probes/patches/, not in csrc/."""
    from pdp_amd import codegen
    retired = ("PDP_F3_SYM_EVERY", "PDP_LQR_UNGUARDED", "PDP_LQR_STREAM_HUX", "PDP_FUSED_CLOSED_LOOP", "PDP_F3_EXP_GAINS_STEP0", "PDP_F3_EXP_IDLE_EVALUATOR", "PDP_LQS_EXP_NOSTORE")
    for f in sorted(os.listdir(codegen.CSRC)):
        if f.endswith((".h", ".hip")):
            src = open(os.path.join(codegen.CSRC, f)).read()
            for name in retired:
                assert name not in src, "%s still mentions %s" % (f, name)
    patches = os.path.join(os.path.dirname(os.path.dirname(codegen.CSRC)), "probes", "patches")
    text = "".join(open(os.path.join(patches, p)).read() for p in os.listdir(patches) if p.endswith(".patch"))
    for name in retired:
        assert name in text, name


def test_bench_solver_model_and_recorded_counter_figures():
    """bench.py's flop / byte model of the OC solve (round-5 verdict, item 1) on the C3 sizes, and the bookkeeping of counter-derived records: the digest of the kernel sources
    is stable, profiles/traffic.json and profiles/latency_floors.json carry one, and every latency-bound entry bench.py maps a floor onto exists in the record."""
    import json
    import bench
    from pdp_amd import codegen
    m = bench.oc_solve_model(13, 4, 50, 2.0, 2.0, 1024, 0.241)
    # 50 stages x (backward 16.9 k + forward 1.0 k) flop per Newton iteration; 9 + 5 big and 10 + 8 small MFMAs per stage executed (13 + 5 and 9 + 8 until round 6)
    assert abs(m["algorithmic_flop_per_solve"] - 2 * 50 * (16900 + 1034)) < 1 and m["executed_mfma_flop_per_solve"] == 2 * 50 * (14 * 2048 + 18 * 512)
    assert m["mfma_issue_cycles_per_iteration"] == 50 * (14 * 64 + 18 * 28)
    assert 0.05 < m["frac_of_fp64_mfma_peak"] < 0.15 and 0.2 < m["frac_of_hbm_peak"] < 0.35
    assert 4.5e5 < m["algorithmic_bytes_per_solve"] < 6e5                 # ~0.5 MB per trajectory: x 1024 = the ~500 MB the counters measured per launch
    small = bench.oc_solve_model(4, 1, 50, 2.2, 2.0, 256, 0.141)
    assert small["executed_mfma_flop_per_solve"] == 2.2 * 50 * 13 * 512
    d = codegen.kernel_sources_digest()
    assert d == codegen.kernel_sources_digest() and len(d) == 40
    root = os.path.dirname(os.path.dirname(os.path.abspath(codegen.CSRC)))
    for rec in ("traffic.json", "latency_floors.json"):
        j = json.load(open(os.path.join(root, "profiles", rec)))
        assert len(j["collected"]["kernel_sources_sha1"]) == 40, rec
    fl = json.load(open(os.path.join(root, "profiles", "latency_floors.json")))
    for w in ("sysid", "cp_poly", "cp_poly_c4", "mlp", "oc_c4", "headline", "solve", "solve_c2"):
        e = fl[w]
        assert 0.0 < e["floor_frac"] < 1.0 and abs(e["floor_frac"] + e["parked_on_waits_frac"] + e["issue_stall_frac"] - 1.0) < 0.05, (w, e["floor_frac"])
