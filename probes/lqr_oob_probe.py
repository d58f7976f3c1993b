"""Root-cause experiment for the out-of-bounds reads of lqr_solve_kernel reported in round 1: run a build variant of
libpdp_hip.so (argv[1]; the PDP_LQR_* hooks live in probes/patches/retired_switches.patch since round 6) on GUARD-BANDED operands: every input, output and the
workspace is a view into one NaN-filled allocation, with NaN words directly before and after it.  A read outside an operand that
reaches the result shows up as NaN / a mismatch against the numpy restatement; all 16 <M, NT> instantiations are exercised."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pdp_amd import runtime as rt
from oracle import pdp_oracle as po
if len(sys.argv) > 1:
    rt.CORE_LIB = os.path.abspath(sys.argv[1])
print("library:", rt.CORE_LIB)
rng = np.random.default_rng(0)
POOL = torch.full((1 << 22,), float("nan"), dtype=torch.float64, device="cuda")
cursor = [0]


def banded(a):
    """copy `a` into the NaN pool with 1..7 NaN words before and directly after it"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    cursor[0] += int(rng.integers(1, 8))
    v = POOL[cursor[0]:cursor[0] + a.size].view(a.shape)
    v.copy_(torch.as_tensor(a, device="cuda"))
    cursor[0] += a.size
    return v


def spd(k, s):
    A = rng.standard_normal((k, k))
    return s * (A @ A.T / k + 0.5 * np.eye(k))


worst, bad = 0.0, 0
ONLY = os.environ.get("PDP_PROBE_ONLY")          # "m,nt" restricts the run to one instantiation
for m in (1, 2, 3, 4):
    for nt in (1, 2, 3, 4):
        if ONLY and ONLY != "%d,%d" % (m, nt):
            continue
        n, T, B = 9 + m, 6, 3
        p = (16 - m) + 16 * (nt - 1) - (3 if nt > 1 else 5)
        F = np.eye(n) + 0.1 * rng.standard_normal((B, T, n, n)); G = 0.3 * rng.standard_normal((B, T, n, m)); E = 0.1 * rng.standard_normal((B, T, n, p))
        Hxx = np.stack([np.stack([spd(n, 1.0) for _ in range(T)]) for _ in range(B)]); Huu = np.stack([np.stack([spd(m, 0.5) for _ in range(T)]) for _ in range(B)])
        Hxu = 0.05 * rng.standard_normal((B, T, n, m)); Hxe, Hue = 0.2 * rng.standard_normal((B, T, n, p)), 0.2 * rng.standard_normal((B, T, m, p))
        hxx = np.stack([spd(n, 1.0) for _ in range(B)]); hxe, X0 = 0.2 * rng.standard_normal((B, n, p)), rng.standard_normal((B, n, p))
        cursor[0] = 0
        POOL.fill_(float("nan"))
        d = {k: banded(v) for k, v in dict(F=F, G=G, E=E, Hxx=Hxx, Huu=Huu, Hxu=Hxu, Hxe=Hxe, Hue=Hue, hxx=hxx, hxe=hxe, X0=X0).items()}
        # outputs and workspace come from the torch allocator: dirty its free blocks too
        junk = [torch.full((int(s),), float("nan"), dtype=torch.float64, device="cuda") for s in (3e5, 1e5, 6e5)]
        del junk
        X, U, Lam, st = rt.lqr_solve(d["F"], d["G"], d["Hxx"], d["Huu"], d["hxx"], d["hxe"], E=d["E"], Hxu=d["Hxu"], Hxe=d["Hxe"], Hue=d["Hue"], X0=d["X0"])
        torch.cuda.synchronize()
        Xn, Un, Ln = X.cpu().numpy(), U.cpu().numpy(), Lam.cpu().numpy()
        err = 0.0
        for b in range(B):
            sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]), [hxx[b]], [hxe[b]], X0[b], T)
            for a, r in ((Xn[b], np.stack(sol["state_traj_opt"])), (Un[b], np.stack(sol["control_traj_opt"])), (Ln[b], np.stack(sol["costate_traj_opt"]))):
                e = np.abs(a - r).max() / max(1.0, np.abs(r).max())
                err = max(err, e if np.isfinite(e) else np.inf)
        flag = "" if err < 1e-10 else "   <-- MISMATCH"
        if err >= 1e-10:
            for b in range(B):
                sol = po.lqr_solver(list(F[b]), list(G[b]), list(E[b]), list(Hxx[b]), list(Huu[b]), list(Hxu[b]), list(Hxe[b]), list(Hue[b]), [hxx[b]], [hxe[b]], X0[b], T)
                for nm, a, r in (("X", Xn[b], np.stack(sol["state_traj_opt"])), ("U", Un[b], np.stack(sol["control_traj_opt"])), ("Lam", Ln[b], np.stack(sol["costate_traj_opt"]))):
                    dd = np.abs(a - r)
                    dd[~np.isfinite(dd)] = np.inf
                    print("     b=%d %s: per-t max err" % (b, nm), ["%.1e" % dd[t].max() for t in range(dd.shape[0])], "nan count", int(np.isnan(a).sum()))
        bad += err >= 1e-10
        worst = max(worst, err)
        print("M=%d NT=%d n=%d p=%d: status %s  max rel err %.2e%s" % (m, nt, n, p, st.cpu().numpy(), err, flag))
print("RESULT %s: %d of 16 instantiations mismatch, worst %.2e" % (os.path.basename(rt.CORE_LIB), bad, worst))
