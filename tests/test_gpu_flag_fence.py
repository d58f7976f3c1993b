"""Fence around the hidden LLVM option -amdgpu-mfma-vgpr-form (codegen.HIP_FLAGS): the headline model libraries (C3 quadrotor, C4 rocket OC units)
are built twice by __graft_entry__.build() - with the option (the shipped build) and with plain -O3 (lib/*__plain.so) - and every kernel that runs MFMA
chains must give BIT-IDENTICAL results in both: the option changes where accumulators live, never the arithmetic.  A miscompile of the kind
profiles/r02_lqr_oob_root_cause.txt describes shows up here as a difference (inputs sit in NaN-dirtied allocations)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(system):
    from pdp_amd import codegen, runtime, zoo
    zoo.register_tuned()
    info = codegen.generate(zoo.make_problem(system, "irl"))[1]
    tuned, plain = codegen.lib_path(info["name"]), codegen.lib_path(info["name"] + "__plain")
    if not codegen.tuned(info["name"]):
        pytest.skip("the option is switched off (PDP_MFMA_VGPR_FORM=0): nothing to fence")
    if not os.path.exists(plain):
        codegen.write_header(zoo.make_problem(system, "irl"))
        codegen.compile_model(info["name"], plain_twin=True)
    return runtime.ModelLib(tuned), runtime.ModelLib(plain)


def _dirty(torch, a):
    """a copy of `a` inside a NaN-filled allocation with NaN words directly around it"""
    a = torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64)
    big = torch.full((a.numel() + 128,), float("nan"), dtype=torch.float64, device="cuda")
    big[64:64 + a.numel()] = a.reshape(-1).cuda()
    return big[64:64 + a.numel()].view(a.shape)


@pytest.mark.parametrize("system,T,B", [("quadrotor", 50, 64), ("rocket", 100, 32)])
def test_tuned_and_plain_builds_agree_bit_for_bit(system, T, B):
    import torch
    import bench
    from pdp_amd import JinEnv
    a, b = _pair(system)
    rng = np.random.default_rng(3)
    if system == "quadrotor":
        x0, u, dx, du = bench.synth_inputs(B, 77)
        u, dx, du = u[:, :T], dx[:, :T + 1], du[:, :T]
        th = np.array(bench.THETA)
    else:
        x0 = np.zeros((B, 13))
        x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
        x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        u = np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3))
        dx, du = np.zeros((B, T + 1, 13)), np.zeros((B, T, 3))
        th = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    args = [_dirty(torch, v) for v in (u, th, dx, du)]
    x0d = _dirty(torch, x0)
    # fused gradient unit (the headline kernel), with sensitivities
    oa = a.oc_pdp_grad(*args, x0=x0d, want_sens=True)
    ob = b.oc_pdp_grad(*args, x0=x0d, want_sens=True)
    for k in ("x", "lam", "loss", "grad", "dxdp", "dudp", "status"):
        assert torch.equal(oa[k], ob[k]), "fused unit, %s: the two builds differ" % k
    assert int(oa["status"].sum()) == 0
    # multiple-shooting OC solve, cold (the whole IPOPT-style iteration: any difference in a Riccati step changes the iterates)
    sa = a.oc_solve_ms(x0d, args[1], T, want_gains=True)
    sb = b.oc_solve_ms(x0d, args[1], T, want_gains=True)
    for k in ("state", "control", "costate", "cost", "iterations", "status", "gains"):
        assert torch.equal(sa[k], sb[k]), "OC solve, %s: the two builds differ" % k
    assert int(sa["converged"].sum()) > 0          # (a cold rocket solve from these perturbed poses need not converge everywhere; equality is the point)
    # materialised kernels of the model library
    lam = oa["lam"]
    xa = a.oc_auxsys(oa["x"], args[0], lam, args[1])
    xb = b.oc_auxsys(oa["x"], args[0], lam, args[1])
    for k in xa:
        assert torch.equal(xa[k], xb[k]), "getAuxSys, %s: the two builds differ" % k
