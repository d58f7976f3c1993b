"""GPU parity: pdp_lqr_solve_batched / aux integrators (HIP, through the C-ABI) vs
  - outputs of the reference's own LQR.lqrSolver (tests/golden/ref_lqr_*.npz), and
  - the numpy oracle on seeded random problems (incl. p > 16 - m tiles, shared/time-invariant strides).
Tolerance: 1e-10 relative to the largest entry of each trajectory (stated fp64 tolerance)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _to_np(t):
    return t.detach().cpu().numpy()


def test_lqr_matches_reference_random_cases(golden_dir):
    from pdp_amd import runtime as rt
    r = np.load(os.path.join(golden_dir, "ref_lqr_random.npz"))
    for c in range(int(r["n_cases"])):
        g = lambda k: r["c%d_%s" % (c, k)]
        T = int(g("T"))
        tv = bool(g("time_varying"))
        sel = (lambda a: a) if tv else (lambda a: a[0])
        X, U, Lam, st = rt.lqr_solve(sel(g("F")), sel(g("G")), sel(g("Hxx")), sel(g("Huu")), g("hxx"), g("hxe"), E=sel(g("E")), Hxu=sel(g("Hxu")),
                                     Hxe=sel(g("Hxe")), Hue=sel(g("Hue")), X0=g("X0"), T=T)
        assert int(st.sum()) == 0
        assert _rel(_to_np(X)[0], g("X")) < TOL, "case %d X" % c
        assert _rel(_to_np(U)[0], g("U")) < TOL, "case %d U" % c
        assert _rel(_to_np(Lam)[0], g("Lam")) < TOL, "case %d Lam" % c


@pytest.mark.parametrize("name", ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
def test_lqr_matches_reference_on_demo_aux_systems(golden_dir, name):
    """aux systems of the stored demos (reference getAuxSys) -> reference lqrSolver outputs, all demos as one batch."""
    from pdp_amd import runtime as rt
    a = np.load(os.path.join(golden_dir, "ref_auxsys_%s.npz" % name))
    l = np.load(os.path.join(golden_dir, "ref_lqr_%s.npz" % name))
    X, U, Lam, st = rt.lqr_solve(a["dynF"], a["dynG"], a["Hxx"], a["Huu"], a["hxx"][:, 0], a["hxe"][:, 0], E=a["dynE"], Hxu=a["Hxu"],
                                 Hxe=a["Hxe"], Hue=a["Hue"])
    assert int(st.sum()) == 0
    assert _rel(_to_np(X), l["X"]) < TOL
    assert _rel(_to_np(U), l["U"]) < TOL
    assert _rel(_to_np(Lam), l["Lam"]) < TOL


@pytest.mark.parametrize("n,m,p,T,B", [(13, 4, 9, 50, 64), (13, 3, 10, 100, 16), (4, 1, 7, 50, 256), (16, 4, 12, 8, 3), (7, 2, 40, 11, 5),
                                        (13, 4, 60, 6, 2), (2, 1, 1, 1, 1), (5, 3, 13, 2, 4), (13, 4, 9, 1, 5), (9, 2, 5, 3, 7), (6, 1, 15, 5, 2)])
def test_lqr_matches_oracle_seeded(n, m, p, T, B):
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(1000 * n + 10 * p + T)

    def spd(k, s):
        A = rng.standard_normal((k, k))
        return s * (A @ A.T / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n))
    G = 0.3 * rng.standard_normal((B, T, n, m))
    E = 0.1 * rng.standard_normal((B, T, n, p))
    Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)])
    Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
    Hxu = 0.05 * rng.standard_normal((B, T, n, m))
    Hxe = 0.2 * rng.standard_normal((B, T, n, p))
    Hue = 0.2 * rng.standard_normal((B, T, m, p))
    hxx = np.stack([spd(n, 1.0) for _ in range(B)])
    hxe = 0.2 * rng.standard_normal((B, n, p))
    X0 = rng.standard_normal((B, n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0)
    assert int(st.sum()) == 0
    X, U, Lam = _to_np(X), _to_np(U), _to_np(Lam)
    for b in range(min(B, 4)):
        sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]),
                            [hxx[b]], [hxe[b]], X0[b], T)
        assert _rel(X[b], np.stack(sol["state_traj_opt"])) < TOL
        assert _rel(U[b], np.stack(sol["control_traj_opt"])) < TOL
        assert _rel(Lam[b], np.stack(sol["costate_traj_opt"])) < TOL


def test_lqr_optional_inputs_and_status():
    """E/Hxu/Hxe/Hue/X0 omitted (zeros, PDP.py:496-555); a singular Huu + G'PG raises the pivot flag."""
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(5)
    n, m, p, T = 6, 2, 3, 9
    F = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    G = rng.standard_normal((n, m))
    Hxx, Huu = np.eye(n), 0.3 * np.eye(m)
    hxx, hxe = np.eye(n), rng.standard_normal((n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, T=T)
    Z = lambda r, c: T * [np.zeros((r, c))]
    sol = po.lqr_solver(T * [F], T * [G], Z(n, p), T * [Hxx], T * [Huu], Z(n, m), Z(n, p), Z(m, p), [hxx], [hxe], np.zeros((n, p)), T)
    assert int(st.sum()) == 0
    assert _rel(_to_np(X)[0], np.stack(sol["state_traj_opt"])) < TOL
    assert _rel(_to_np(U)[0], np.stack(sol["control_traj_opt"])) < TOL
    X, U, Lam, st = rt.lqr_solve(F, np.zeros((n, m)), Hxx, np.zeros((m, m)), hxx, hxe, T=T)
    assert int(st[0]) & 2


def test_aux_integrators_match_numpy():
    from pdp_amd import runtime as rt
    rng = np.random.default_rng(11)
    B, T, n, m, p = 3, 7, 13, 4, 37
    F = rng.standard_normal((B, T, n, n)) * 0.3
    G = rng.standard_normal((B, T, n, m))
    Ux = rng.standard_normal((B, T, m, n)) * 0.2
    Ue = rng.standard_normal((B, T, m, p))
    E = rng.standard_normal((B, T, n, p))
    X0 = rng.standard_normal((B, n, p))
    X, U = rt.cp_aux_integrate(F, G, Ux, Ue, X0)
    Xs = rt.sysid_aux_integrate(F, E, None)
    X, U, Xs = _to_np(X), _to_np(U), _to_np(Xs)
    for b in range(B):
        x = X0[b]
        xs = np.zeros((n, p))
        for t in range(T):
            u = Ux[b, t] @ x + Ue[b, t]
            x = F[b, t] @ x + G[b, t] @ u
            xs = F[b, t] @ xs + E[b, t]
            assert _rel(U[b, t], u) < 1e-12 and _rel(X[b, t + 1], x) < 1e-12 and _rel(Xs[b, t + 1], xs) < 1e-12


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("nt", [1, 2, 3, 4])
def test_lqr_every_kernel_instantiation(m, nt):
    """all 16 instantiations lqr_solve_kernel<M, NT> (control dimension x number of 16-column parameter tiles), on memory the
    allocator has handed out before (NaN-filled), against the numpy restatement of PDP.py:557-608"""
    import torch
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    n, T, B = 9 + m, 6, 3
    p = (16 - m) + 16 * (nt - 1) - (3 if nt > 1 else 5)          # last tile partly filled
    rng = np.random.default_rng(100 * m + nt)
    junk = [torch.full((int(s),), float("nan"), dtype=torch.float64, device="cuda") for s in (2e5, 7e5, 1e5)]
    del junk

    def spd(k, s):
        A = rng.standard_normal((k, k))
        return s * (A @ A.T / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n))
    G = 0.3 * rng.standard_normal((B, T, n, m))
    E = 0.1 * rng.standard_normal((B, T, n, p))
    Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)])
    Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
    Hxu = 0.05 * rng.standard_normal((B, T, n, m))
    Hxe, Hue = 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
    hxx = np.stack([spd(n, 1.0) for _ in range(B)])
    hxe, X0 = 0.2 * rng.standard_normal((B, n, p)), rng.standard_normal((B, n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0)
    assert int(st.sum()) == 0
    X, U, Lam = _to_np(X), _to_np(U), _to_np(Lam)
    for b in range(B):
        sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]),
                            [hxx[b]], [hxe[b]], X0[b], T)
        assert _rel(X[b], np.stack(sol["state_traj_opt"])) < TOL and _rel(U[b], np.stack(sol["control_traj_opt"])) < TOL
        assert _rel(Lam[b], np.stack(sol["costate_traj_opt"])) < TOL


@pytest.mark.parametrize("n,m,p,B", [(4, 1, 7, 9), (2, 1, 5, 4), (4, 2, 8, 6), (3, 3, 13, 5), (4, 4, 12, 7), (4, 1, 15, 2), (1, 1, 1, 1)])
def test_lqr_small_systems_four_trajectories_per_wavefront(n, m, p, B):
    """n <= 4, m + p <= 16: lqr_solve_small_kernel<M> packs four trajectories block-diagonally into one tile (pdp_riccati_small.h).
    Every M, batches that do not fill the last wavefront, full / minimal parameter widths, on NaN-dirtied memory, against the numpy
    restatement of PDP.py:557-608; then the optional-input and broadcast (time-invariant, shared-over-batch) forms."""
    import torch
    from oracle import pdp_oracle as po
    from pdp_amd import runtime as rt
    T = 11
    rng = np.random.default_rng(1000 * n + 100 * m + p)
    junk = [torch.full((int(s),), float("nan"), dtype=torch.float64, device="cuda") for s in (2e5, 7e5, 1e5)]
    del junk

    def spd(k, s):
        A = rng.standard_normal((k, k))
        return s * (A @ A.T / k + 0.5 * np.eye(k))
    F = np.eye(n) + 0.2 * rng.standard_normal((B, T, n, n))
    G = 0.5 * rng.standard_normal((B, T, n, m))
    E = 0.1 * rng.standard_normal((B, T, n, p))
    Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)])
    Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
    Hxu = 0.05 * rng.standard_normal((B, T, n, m))
    Hxe, Hue = 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
    hxx = np.stack([spd(n, 1.0) for _ in range(B)])
    hxe, X0 = 0.2 * rng.standard_normal((B, n, p)), rng.standard_normal((B, n, p))
    X, U, Lam, st = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0)
    assert int(st.sum()) == 0
    X, U, Lam = _to_np(X), _to_np(U), _to_np(Lam)
    for b in range(B):
        sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]),
                            [hxx[b]], [hxe[b]], X0[b], T)
        assert _rel(X[b], np.stack(sol["state_traj_opt"])) < TOL and _rel(U[b], np.stack(sol["control_traj_opt"])) < TOL
        assert _rel(Lam[b], np.stack(sol["costate_traj_opt"])) < TOL
    # without the costate output; optional families omitted; time-invariant matrices shared over the batch
    X2, U2, L2, _ = rt.lqr_solve(F, G, Hxx, Huu, hxx, hxe, E=E, Hxu=Hxu, Hxe=Hxe, Hue=Hue, X0=X0, want_costate=False)
    assert L2 is None and np.array_equal(_to_np(X2), X) and np.array_equal(_to_np(U2), U)
    X3, U3, _, st3 = rt.lqr_solve(F[0, 0], G[0, 0], Hxx[0, 0], Huu[0, 0], hxx, hxe, T=T)
    Z = lambda r, c: T * [np.zeros((r, c))]
    for b in (0, B - 1):
        sol = po.lqr_solver(T * [F[0, 0]], T * [G[0, 0]], Z(n, p), T * [Hxx[0, 0]], T * [Huu[0, 0]], Z(n, m), Z(n, p), Z(m, p), [hxx[b]], [hxe[b]], np.zeros((n, p)), T)
        assert _rel(_to_np(X3)[b], np.stack(sol["state_traj_opt"])) < TOL and _rel(_to_np(U3)[b], np.stack(sol["control_traj_opt"])) < TOL
    # a singular control block raises the pivot flag of that trajectory only
    Gz, Huz = G.copy(), Huu.copy()
    Gz[B - 1], Huz[B - 1] = 0.0, 0.0
    _, _, _, st4 = rt.lqr_solve(F, Gz, Hxx, Huz, hxx, hxe, E=E, Hxe=Hxe, Hue=Hue)
    st4 = _to_np(st4)
    assert st4[B - 1] & 2 and (B == 1 or int(st4[:B - 1].sum()) == 0)


def test_lqr_guard_banded_operands_every_instantiation():
    """Bounds check by construction (probes/lqr_oob_probe.py): every operand of all 16 lqr_solve_kernel<M, NT> instantiations is a view
    into ONE NaN-filled allocation with NaN words directly before and after it, outputs and workspace come from NaN-dirtied allocator
    blocks; a read outside an operand that reaches the result is a NaN / mismatch against the numpy restatement.  (Round 1's unguarded
    prefetch after the last step, and a miscompiled streamed-operand variant, both fail this probe: DESIGN.md section 8.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-u", os.path.join(root, "probes", "lqr_oob_probe.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT libpdp_hip.so: 0 of 16 instantiations mismatch" in r.stdout, r.stdout[-3000:]


def test_one_wave_lqr_kernel_stays_parity_green():
    """The runner / streamer kernel (lqr_solve_stream_kernel) is the default wherever it applies; the one-wave kernel behind it (more than one
    parameter tile, the single-shooting solver's LQ step, PDP_LQR_VARIANT=1) must keep producing the same results: this file's reference /
    oracle tests rerun in a process with the variant selected."""
    import subprocess, sys
    env = dict(os.environ, PDP_LQR_VARIANT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-k",
                        "matches_reference or matches_oracle_seeded or optional_inputs"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:]
