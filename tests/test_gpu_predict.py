"""GPU: the sensitivity-predicted starting point of an IRL loop's next OC solve (round 4).

The auxiliary control system of PDP IS the derivative of the optimal trajectory with respect to theta (reference PDP/PDP.py:272-314 getAuxSys,
557-608 lqrSolver: X = dx/dtheta, U = du/dtheta, Lambda_t = P_{t+1} X_{t+1} + W_{t+1}).  The gradient unit keeps it on request
(pdp_oc_pdp_grad_sens_batched: dxdp, dudp and the Riccati record P_{t+1} | W_{t+1}), pdp_oc_predict_batched applies it to a parameter step, and the
multiple-shooting solver started there (PDP_MS_WARM) needs one Newton iteration fewer than from the previous solution.  Checked here against the oracle:
the record against lqr_solver's PP / WW, the predicted point against ipopt_ms.predict_start, the solve from it against ipopt_ms.solve(warm=...) row by row."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def setup(golden_dir, name):
    from oracle import models, pdp_oracle as po
    from pdp_amd import zoo
    d = np.load(os.path.join(golden_dir, "demos_%s.npz" % name))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    return d, oc, zoo.get(name, "irl")


def npy(t):
    return t.detach().cpu().numpy()


def solve_at(mdl, x0, th, T):
    sol = mdl.oc_solve_ms(x0, th, T, tol=1e-11)
    assert bool(sol["converged"].all())
    return sol


@pytest.mark.parametrize("name", ["pendulum", "cartpole", "robotarm", "quadrotor", "rocket"])
def test_riccati_record_and_predicted_point_match_the_oracle(golden_dir, margins, name):
    """at the optimum of the stored demos' problems: the Riccati record the gradient unit leaves is PP[t], WW[t] of the reference's lqrSolver on the aux system
    of that point, and (x, u, lam) + (X, U, Lambda) dtheta from pdp_oc_predict_batched is the oracle's predict_start - with per-sample and shared dtheta, with and
    without the multiplier part; the gradient itself is unchanged by asking for the record."""
    from oracle import ipopt_ms, pdp_oracle as po
    d, oc, mdl = setup(golden_dir, name)
    th = d["true_parameter"]
    B, T = d["state"].shape[0], d["control"].shape[1]
    sol = solve_at(mdl, d["state"][:, 0], th, T)
    out = mdl.oc_pdp_grad(sol["control"], th, d["state"], d["control"], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True)
    ref = mdl.oc_pdp_grad(sol["control"], th, d["state"], d["control"], x=sol["state"], lam=sol["costate"], want_sens=True)
    assert int(out["status"].sum()) == 0
    for k in ("loss", "grad", "dxdp", "dudp"):
        assert bool((out[k] == ref[k]).all()), k                       # same arithmetic: the record is a by-product
    n, p = oc.n, oc.p
    ric = out["riccati"].cpu().numpy()
    assert ric.shape == (B, T, n * n + n * p + 1)
    rng = np.random.default_rng(3)
    dth = th[None] * 0.02 * rng.uniform(-1, 1, (B, p))
    xs, us, ls = (sol[k].cpu().numpy() for k in ("state", "control", "costate"))
    xp, up, lp = (a.cpu().numpy() for a in mdl.oc_predict(sol["state"], sol["control"], sol["costate"], dth, out["dxdp"], out["dudp"], out["riccati"]))
    xq, uq, lq = (a.cpu().numpy() for a in mdl.oc_predict(sol["state"], sol["control"], sol["costate"], dth, out["dxdp"], out["dudp"]))
    assert (lq == ls).all() and (xq == xp).all() and (uq == up).all()     # without the record: multipliers untouched, the rest identical
    xs1, us1, ls1 = (a.cpu().numpy() for a in mdl.oc_predict(sol["state"], sol["control"], sol["costate"], dth[0], out["dxdp"], out["dudp"], out["riccati"]))
    sc = lambda a: max(1.0, np.abs(a).max())
    for i in range(min(B, 3)):
        aux = oc.getAuxSys(xs[i], us[i], ls[i], th)
        lqr = po.lqr_from_aux(aux, n, p, T)
        PP, WW = np.stack(lqr["PP"]), np.stack(lqr["WW"])
        margins.check("Riccati record of the gradient unit vs lqr_solver PP, %s sample %d (relative to the largest entry)" % (name, i),
                      np.abs(ric[i, :, :n * n].reshape(T, n, n) - PP).max() / sc(PP), 1e-9)
        margins.check("Riccati record of the gradient unit vs lqr_solver WW, %s sample %d" % (name, i),
                      np.abs(ric[i, :, n * n:n * n + n * p].reshape(T, n, p) - WW).max() / sc(WW), 1e-9)
        ex, eu, el = ipopt_ms.predict_start(oc, xs[i], us[i], ls[i], th, dth[i])
        margins.check("predicted state vs oracle predict_start, %s sample %d" % (name, i), np.abs(xp[i] - ex).max() / sc(ex), 1e-10)
        margins.check("predicted control vs oracle predict_start, %s sample %d" % (name, i), np.abs(up[i] - eu).max() / sc(eu), 1e-10)
        margins.check("predicted costate vs oracle predict_start, %s sample %d" % (name, i), np.abs(lp[i] - el).max() / sc(el), 1e-9)
    ex, eu, el = ipopt_ms.predict_start(oc, xs[B - 1], us[B - 1], ls[B - 1], th, dth[0])      # shared dtheta (stride 0) reaches the last sample too
    assert np.abs(xs1[B - 1] - ex).max() <= 1e-10 * sc(ex) and np.abs(ls1[B - 1] - el).max() <= 1e-9 * sc(el) and np.abs(us1[B - 1] - eu).max() <= 1e-10 * sc(eu)
    assert (xp[:, 0] == xs[:, 0]).all()                                    # x_0 is fixed: X_0 = 0


@pytest.mark.parametrize("name,rel", [("cartpole", 0.05), ("quadrotor", 0.02), ("rocket", 0.02)])
def test_ms_kernel_from_the_predicted_start_follows_the_oracle(golden_dir, name, rel):
    """an IRL step: solve at theta, predict to theta' = theta (1 +- rel), solve at theta' from the predicted point.  The kernel's iteration log from that
    start equals the restatement's (ipopt_ms.solve(warm=predict_start(...))) row by row, it ends in the same optimum as a plain warm start - and takes
    fewer iterations than that."""
    from oracle import ipopt_ms
    d, oc, mdl = setup(golden_dir, name)
    th = d["true_parameter"]
    T = d["control"].shape[1]
    x0 = d["state"][:1, 0]
    sol = solve_at(mdl, x0, th, T)
    out = mdl.oc_pdp_grad(sol["control"], th, d["state"][:1], d["control"][:1], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True)
    rng = np.random.default_rng(7)
    th1 = th * (1 + rel * rng.uniform(-1, 1, th.shape))
    pred = mdl.oc_predict(sol["state"], sol["control"], sol["costate"], th1 - th, out["dxdp"], out["dudp"], out["riccati"])
    xs, us, ls = (sol[k][0].cpu().numpy() for k in ("state", "control", "costate"))
    log = []
    ref = ipopt_ms.solve(oc, x0[0], T, th1, tol=1e-10, log=log, warm=ipopt_ms.predict_start(oc, xs, us, ls, th, th1 - th))
    got = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=pred, log_rows=len(log) + 4)
    # the same start applied INSIDE the solver launch (PDP_MS_PREDICT) from the previous solution: the same arithmetic in the same order - identical iterates
    fused = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]), log_rows=len(log) + 4,
                            predict=dict(dtheta=th1 - th, dxdp=out["dxdp"], dudp=out["dudp"], riccati=out["riccati"]))
    for k in ("state", "control", "costate", "cost", "iterations", "log"):
        assert bool((fused[k] == got[k]).all()), k
    plain = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]))
    assert bool(got["converged"][0]) and int(got["status"][0]) == 0 and bool(plain["converged"][0])
    assert int(got["iterations"][0]) == ref["iterations"] == len(log)
    assert int(got["iterations"][0]) < int(plain["iterations"][0])
    kl = got["log"][0].cpu().numpy()
    for r, l in zip(kl, log):
        assert r[5] == (-l["alpha"] if l["soc_taken"] else l["alpha"]) and r[4] == l["dw"], (name, l["it"], r[4], r[5], l["dw"], l["alpha"])
        assert abs(r[1] - l["f"]) <= 1e-9 * max(1.0, abs(l["f"])) and abs(r[2] - l["inf_pr"]) <= 1e-8 * max(1e-3, l["inf_pr"])
    sc = lambda a: max(1.0, np.abs(a).max())
    for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
        assert np.abs(got[k][0].cpu().numpy() - ref[kr]).max() <= 1e-9 * sc(ref[kr])
        assert float((got[k] - plain[k]).abs().max()) <= 1e-8 * sc(ref[kr])


def test_predicted_start_cuts_the_iterations_of_a_batch(golden_dir):
    """BASELINE config C2's shape: 256 cart-pole problems, per-sample parameters theta* +- 0.05.  From the predicted point the slowest trajectory of the batch
    (the launch lasts as long as it does) needs fewer iterations than from the solution at theta*, and every sample lands in the same optimum."""
    _, _, mdl = setup(golden_dir, "cartpole")
    rng = np.random.default_rng(0)
    B, T = 256, 50
    th = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
    x0 = np.zeros((B, 4))
    x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    th1 = th[None] + rng.uniform(-0.05, 0.05, (B, 7))
    sol = solve_at(mdl, x0, th, T)
    demo = (sol["state"], sol["control"])
    out = mdl.oc_pdp_grad(sol["control"], th, demo[0], demo[1], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True)
    pred = mdl.oc_predict(sol["state"], sol["control"], sol["costate"], th1 - th[None], out["dxdp"], out["dudp"], out["riccati"])
    a = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]))
    b = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=pred)
    # the prediction applied inside the solver launch (PDP_MS_PREDICT) from the previous solution: the same arithmetic as the predict kernel - identical results
    c = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]),
                        predict=dict(dtheta=th1 - th[None], dxdp=out["dxdp"], dudp=out["dudp"], riccati=out["riccati"]))
    assert bool((c["iterations"] == b["iterations"]).all()) and bool((c["state"] == b["state"]).all()) and bool((c["costate"] == b["costate"]).all())
    # states and controls only (no Riccati record): multipliers start where they were
    pxu = mdl.oc_predict(sol["state"], sol["control"], sol["costate"], th1 - th[None], out["dxdp"], out["dudp"])
    e = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=pxu)
    f = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]), predict=dict(dtheta=th1 - th[None], dxdp=out["dxdp"], dudp=out["dudp"], guard=False))
    assert bool(e["converged"].all()) and bool((e["iterations"] == f["iterations"]).all()) and bool((e["state"] == f["state"]).all())
    # ... and guarded (the default): at steps of up to 10 % a few of the 256 primal-only predictions are worse in the constraints than the point they started from and are
    # dropped (status & 512); those samples run the plain warm solve `a`, the others the predicted one - bit for bit
    fg = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]), predict=dict(dtheta=th1 - th[None], dxdp=out["dxdp"], dudp=out["dudp"]))
    rej = (fg["status"] & 512) != 0
    assert bool(fg["converged"].all()) and 0 < int(rej.sum()) < B // 4
    assert bool((fg["state"][~rej] == f["state"][~rej]).all()) and bool((fg["iterations"][~rej] == f["iterations"][~rej]).all())
    assert bool(a["converged"].all()) and bool(b["converged"].all())
    assert bool((fg["state"][rej] == a["state"][rej]).all()) and bool((fg["iterations"][rej] == a["iterations"][rej]).all())
    ia, ib = a["iterations"].double(), b["iterations"].double()
    print("iterations from the previous solution: mean %.2f max %d; from the predicted point: mean %.2f max %d" % (float(ia.mean()), int(ia.max()), float(ib.mean()), int(ib.max())))
    assert float(ib.mean()) <= float(ia.mean()) - 0.5 and int(ib.max()) < int(ia.max())
    for k in ("state", "control", "costate"):
        assert float((a[k] - b[k]).abs().max()) <= 1e-7 * max(1.0, float(a[k].abs().max()))


WORKER = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
from pdp_amd import zoo
out = {}
for name, B in (("cartpole", 9), ("quadrotor", 2)):
    d = np.load(%(golden)r + "/demos_" + name + ".npz")
    mdl = zoo.get(name, "irl")
    th, T = d["true_parameter"], d["control"].shape[1]
    x0 = np.repeat(d["state"][:1, 0], B, axis=0)
    x0[:, 1] += 0.01 * np.arange(B)
    sol = mdl.oc_solve_ms(x0, th, T, tol=1e-11)
    o = mdl.oc_pdp_grad(sol["control"], th, sol["state"], sol["control"], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True)
    th1 = th[None] * (1 + 0.02 * np.random.default_rng(1).uniform(-1, 1, (B, th.size)))
    r = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]),
                        predict=dict(dtheta=th1 - th[None], dxdp=o["dxdp"], dudp=o["dudp"], riccati=o["riccati"]))
    assert bool(r["converged"].all())
    for k in ("state", "control", "costate", "iterations"):
        out[name + "_" + k] = r[k].cpu().numpy()
    o2 = mdl.oc_pdp_grad(sol["control"], th, sol["state"], sol["control"], x=sol["state"], lam=sol["costate"], want_predict_record=True)
    r2 = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(sol["state"], sol["control"], sol["costate"]), predict=dict(dtheta=th1 - th[None], record=o2["predict_record"]))
    assert bool(r2["converged"].all())
    for k in ("state", "control", "costate", "iterations"):
        out[name + "_rec_" + k] = r2[k].cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_prediction_in_front_of_the_one_wave_kernel(golden_dir, tmp_path):
    """PDP_MS_VARIANT=1 (the one-wave solver kernel) takes PDP_MS_PREDICT as a launch of pdp_oc_predict_batched in front of it: same iterations, same optimum as the
    runner / evaluator kernel, which applies the prediction while loading the point."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for variant in ("2", "1"):
        path = str(tmp_path / ("v%s.npz" % variant))
        r = subprocess.run([sys.executable, "-c", WORKER % dict(root=root, golden=golden_dir), path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900,
                           env=dict(os.environ, PDP_MS_VARIANT=variant))
        assert r.returncode == 0, r.stdout[-3000:]
        res[variant] = np.load(path)
    for k in res["2"].files:
        a, b = res["2"][k], res["1"][k]
        if k.endswith("iterations"):
            assert np.array_equal(a, b), k
        else:
            assert np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(a).max()), k


@pytest.mark.parametrize("name,B", [("cartpole", 5), ("quadrotor", 2), ("rocket", 2)])
def test_packed_fp32_prediction_record(golden_dir, margins, name, B):
    """the single-precision record an IRL loop passes from the gradient unit to the next solve (pdp_oc_sens_out.predict_record: X_{t+1} | U_t | triangle of P_{t+1} |
    W_{t+1} per stage): its entries are the fp64 outputs rounded to fp32; the point predicted from it agrees with the oracle's to fp32 accuracy OF THE CORRECTION
    (1e-6 of the step); the solver started from it - inside the launch (PDP_MS_PREDICT with the record) or from pdp_oc_predict_record_batched - takes the iterations
    the oracle takes from its own (fp64) prediction and lands in the same optimum; asking for the record changes nothing in loss and gradient."""
    from oracle import ipopt_ms
    d, oc, mdl = setup(golden_dir, name)
    th = d["true_parameter"]
    T = d["control"].shape[1]
    n, m, p = oc.n, oc.m, oc.p
    x0 = np.repeat(d["state"][:1, 0], B, axis=0)
    x0[:, 1] += 0.01 * np.arange(B)
    sol = solve_at(mdl, x0, th, T)
    dem = (d["state"][:1].repeat(B, axis=0), d["control"][:1].repeat(B, axis=0))
    a = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True, want_predict_record=True)
    b = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"])
    c = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"], want_predict_record=True)      # the record alone
    assert bool((a["loss"] == b["loss"]).all()) and bool((a["grad"] == b["grad"]).all()) and bool((c["grad"] == b["grad"]).all())
    rec = a["predict_record"].cpu().numpy()
    assert rec.dtype == np.float32 and rec.shape == (B, T, 2 * n * p + m * p + n * (n + 1) // 2) and np.array_equal(rec, c["predict_record"].cpu().numpy())
    X, U, R = a["dxdp"].cpu().numpy(), a["dudp"].cpu().numpy(), a["riccati"].cpu().numpy()
    P, W = R[:, :, :n * n].reshape(B, T, n, n), R[:, :, n * n:n * n + n * p].reshape(B, T, n, p)
    iu = np.triu_indices(n)
    want = np.concatenate([X[:, 1:].reshape(B, T, -1), U.reshape(B, T, -1), P[:, :, iu[0], iu[1]], W.reshape(B, T, -1)], axis=2).astype(np.float32)
    assert np.array_equal(rec, want)
    rng = np.random.default_rng(5)
    th1 = th[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, p)))
    dth = th1 - th[None]
    xs, us, ls = (sol[k].cpu().numpy() for k in ("state", "control", "costate"))
    xp, up, lp = (t_.cpu().numpy() for t_ in mdl.oc_predict_from_record(sol["state"], sol["control"], sol["costate"], dth, a["predict_record"]))
    warm = (sol["state"], sol["control"], sol["costate"])
    inl = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, record=a["predict_record"]))
    pre = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(xp, up, lp))
    assert bool(inl["converged"].all()) and bool(pre["converged"].all())
    for i in range(B):
        ex, eu, el = ipopt_ms.predict_start(oc, xs[i], us[i], ls[i], th, dth[i])
        for lab, got, e, base in (("state", xp[i], ex, xs[i]), ("control", up[i], eu, us[i]), ("costate", lp[i], el, ls[i])):
            margins.check("point predicted from the fp32 record vs oracle, %s sample %d: %s (relative to the size of the correction)" % (name, i, lab),
                          np.abs(got - e).max() / max(1e-300, np.abs(e - base).max()), 1e-5 if lab == "costate" else 1e-6)      # (P dx and W dtheta cancel in part)
        ref = ipopt_ms.solve(oc, x0[i], T, th1[i], tol=1e-10, warm=(ex, eu, el))
        assert int(inl["iterations"][i]) == ref["iterations"] == int(pre["iterations"][i]), (name, i)
        for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
            sc_ = max(1.0, np.abs(ref[kr]).max())
            assert np.abs(inl[k][i].cpu().numpy() - ref[kr]).max() <= 1e-9 * sc_ and np.abs(pre[k][i].cpu().numpy() - ref[kr]).max() <= 1e-9 * sc_


@pytest.mark.parametrize("name,B", [("cartpole", 5), ("quadrotor", 3)])
def test_primal_prediction_record(golden_dir, name, B):
    """PDP_OC_RECORD_PRIMAL / PDP_MS_PREDICT_PRIMAL: the gradient unit writes only the X | U part of the record (bit for bit the full record's, the rest of the
    buffer is left as it was), the solver started from it inside the launch starts where pdp_oc_predict_record_batched without multipliers puts it (same
    iterations, same answer as the oracle from that point), and loss and gradient do not depend on the kind of record asked for."""
    import torch
    from oracle import ipopt_ms
    d, oc, mdl = setup(golden_dir, name)
    th = d["true_parameter"]
    T = d["control"].shape[1]
    n, m, p = oc.n, oc.m, oc.p
    x0 = np.repeat(d["state"][:1, 0], B, axis=0)
    x0[:, 1] += 0.01 * np.arange(B)
    sol = solve_at(mdl, x0, th, T)
    dem = (d["state"][:1].repeat(B, axis=0), d["control"][:1].repeat(B, axis=0))
    full = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"], want_predict_record=True)
    nrec = full["predict_record"].shape[2]
    mark = torch.full((B, T, nrec), -7.0, dtype=torch.float32, device="cuda")
    prim = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"], want_predict_record="primal", buffers={"predict_record": mark})
    assert bool((prim["loss"] == full["loss"]).all()) and bool((prim["grad"] == full["grad"]).all())
    rp, rf = prim["predict_record"].cpu().numpy(), full["predict_record"].cpu().numpy()
    nxu = (n + m) * p
    assert np.array_equal(rp[:, :, :nxu], rf[:, :, :nxu]) and (rp[:, :, nxu:] == -7.0).all()
    rng = np.random.default_rng(11)
    th1 = th[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, p)))
    dth = th1 - th[None]
    xp, up, lp = mdl.oc_predict_from_record(sol["state"], sol["control"], sol["costate"], dth, prim["predict_record"], primal=True)
    xf, uf, _ = mdl.oc_predict_from_record(sol["state"], sol["control"], sol["costate"], dth, full["predict_record"])
    assert bool((xp == xf).all()) and bool((up == uf).all()) and bool((lp == sol["costate"]).all())
    warm = (sol["state"], sol["control"], sol["costate"])
    inl = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, record=prim["predict_record"], primal=True), log_rows=8)
    pre = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(xp, up, lp), log_rows=8)
    assert bool(inl["converged"].all()) and bool((inl["iterations"] == pre["iterations"]).all())
    for k in ("state", "control", "costate", "cost"):
        assert bool((inl[k] == pre[k]).all()), k                  # the same starting point, bit for bit: the same iterates
    for i in range(min(B, 2)):
        ref = ipopt_ms.solve(oc, x0[i], T, th1[i], tol=1e-10, warm=(xp[i].cpu().numpy(), up[i].cpu().numpy(), lp[i].cpu().numpy()))
        assert int(inl["iterations"][i]) == ref["iterations"], (name, i)
        for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
            assert np.abs(inl[k][i].cpu().numpy() - ref[kr]).max() <= 1e-9 * max(1.0, np.abs(ref[kr]).max())
    # the flag without a record, or without PDP_MS_PREDICT, is an argument error
    with pytest.raises(RuntimeError):
        mdl.oc_solve_ms(x0, th1, T, warm=warm, predict=dict(dtheta=dth, dxdp=torch.zeros(B, T + 1, n, p, dtype=torch.float64, device="cuda"),
                                                           dudp=torch.zeros(B, T, m, p, dtype=torch.float64, device="cuda"), primal=True))


def test_record_prediction_with_more_controls_than_states():
    """n = 2, m = 3 (round-4 advice): the in-launch prediction from the packed record maps lanes to (stage, state row) and used to load the control rows under `i < NU`
    inside that mapping - rows >= NX were never written and the solver started from uninitialised LDS.  The point the solver starts from inside the launch must be the
    point pdp_oc_predict_record_batched computes (which has always handled m > n): same iterates bit for bit, full and primal record."""
    from pdp_amd import PDP
    from pdp_amd.sx import SX, dot, mtimes, sin
    rng = np.random.default_rng(8)
    n, m, T, B = 2, 3, 20, 6
    A = np.eye(n) + 0.1 * rng.standard_normal((n, n))
    Bm = 0.5 * rng.standard_normal((n, m))
    X, U, w = SX.sym("x", n), SX.sym("u", m), SX.sym("w", 3)
    oc = PDP.OCSys("more controls than states")
    oc.setAuxvarVariable(w)
    oc.setStateVariable(X)
    oc.setControlVariable(U)
    oc.setDyn(mtimes(SX(A), X) + mtimes(SX(Bm), U) + 0.05 * w[2] * sin(X))
    oc.setPathCost(w[0] * dot(X, X) + w[1] * dot(U, U))
    oc.setFinalCost(3.0 * w[0] * dot(X, X))
    mdl = oc.model()
    assert (mdl.n, mdl.m, mdl.p) == (2, 3, 3)
    th = np.array([1.3, 0.4, 1.0])
    x0 = rng.standard_normal((B, n))
    sol = mdl.oc_solve_ms(x0, th, T, tol=1e-10)
    assert bool(sol["converged"].all())
    dem_x, dem_u = np.zeros((B, T + 1, n)), np.zeros((B, T, m))
    th1 = th[None] * (1 + 0.05 * rng.uniform(-1, 1, (B, 3)))
    dth = th1 - th[None]
    warm = (sol["state"], sol["control"], sol["costate"])
    for primal in (False, True):
        g = mdl.oc_pdp_grad(sol["control"], th, dem_x, dem_u, x=sol["state"], lam=sol["costate"], want_predict_record="primal" if primal else True)
        xp, up, lp = mdl.oc_predict_from_record(sol["state"], sol["control"], sol["costate"], dth, g["predict_record"], primal=primal)
        assert float((up - sol["control"]).abs().max()) > 0            # every control row moves
        assert bool(((up - sol["control"]).abs().amax(dim=(0, 1)) > 0).all())
        inl = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, record=g["predict_record"], primal=primal), log_rows=6)
        pre = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(xp, up, lp), log_rows=6)
        assert bool(inl["converged"].all()) and bool((inl["iterations"] == pre["iterations"]).all())
        for k in ("state", "control", "costate", "cost"):
            assert bool((inl[k] == pre[k]).all()), (primal, k)


def test_prediction_guard_on_the_stored_rocket_trace(golden_dir):
    """PDP_MS_PREDICT_GUARD on the reference's own rocket IRL run, parameter step row 0 -> row 1 (tests/golden/irltrace_head_rocket.npz): the first-order prediction has
    ~50 times the KKT error of the previous solution.  Guarded (the default of oc_solve_ms(predict=...)): the kernel drops the prediction (status & 512), starts from the
    previous solution and takes the iterations the oracle takes from there to the optimum IPOPT stored (loss_trace[2] = 1301.2367).  Unguarded: it follows the oracle from
    the predicted point into the other stationary point (loss 10289.857) - the kernel is faithful either way, the guard is what keeps the LOOP on the reference's trace."""
    from oracle import ipopt_ms
    d, oc, mdl = setup(golden_dir, "rocket")
    h = np.load(os.path.join(golden_dir, "irltrace_head_rocket.npz"))
    T, x0 = d["control"].shape[1], d["state"][:1, 0]
    th0, th1 = h["param"][0], h["param"][1]
    sol = mdl.oc_solve_ms(x0, th0, T, tol=1e-10)
    assert bool(sol["converged"].all())
    g = mdl.oc_pdp_grad(sol["control"], th0, d["state"][:1], d["control"][:1], x=sol["state"], lam=sol["costate"], want_predict_record=True)
    warm = (sol["state"], sol["control"], sol["costate"])
    xs, us, ls = (sol[k][0].cpu().numpy() for k in ("state", "control", "costate"))
    start, rejected = ipopt_ms.guarded_start(oc, x0[0], xs, us, ls, th0, th1 - th0)
    assert rejected
    ref = ipopt_ms.solve(oc, x0[0], T, th1, tol=1e-10, warm=start)
    ref_pred = ipopt_ms.solve(oc, x0[0], T, th1, tol=1e-10, warm=ipopt_ms.predict_start(oc, xs, us, ls, th0, th1 - th0))

    def loss_of(x, u):
        return float(np.linalg.norm(x - d["state"][0]) ** 2 + np.linalg.norm(u - d["control"][0]) ** 2)
    for guard, want in ((True, ref), (False, ref_pred)):
        s = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=th1 - th0, record=g["predict_record"], guard=guard))
        assert bool(s["converged"].all())
        assert (int(s["status"][0]) & 512 != 0) == guard
        if guard:       # (from the bad point both take 200+ iterations through inertia corrections and restorations: the same stationary point, not the same path)
            assert int(s["iterations"][0]) == want["iterations"], (int(s["iterations"][0]), want["iterations"])
        for k, kr in (("state", "state_traj_opt"), ("control", "control_traj_opt"), ("costate", "costate_traj_opt")):
            assert np.abs(s[k][0].cpu().numpy() - want[kr]).max() <= (1e-8 if guard else 1e-6) * max(1.0, np.abs(want[kr]).max()), (guard, k)
    assert abs(loss_of(ref_pred["state_traj_opt"], ref_pred["control_traj_opt"]) - 10289.857357) <= 1e-3
    for primal in (True, False):        # the primal-only record is judged by the primal residual alone: rejected here as well
        s = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=th1 - th0, record=g["predict_record"], primal=primal))
        assert int(s["status"][0]) & 512 and abs(loss_of(s["state"][0].cpu().numpy(), s["control"][0].cpu().numpy()) - h["loss"][2]) <= 1e-9 * h["loss"][2]
    guarded = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=th1 - th0, record=g["predict_record"]))
    assert abs(loss_of(guarded["state"][0].cpu().numpy(), guarded["control"][0].cpu().numpy()) - h["loss"][2]) <= 1e-9 * h["loss"][2]


@pytest.mark.parametrize("name,B", [("cartpole", 5), ("quadrotor", 3)])
def test_prediction_guard_changes_nothing_where_the_prediction_is_good(golden_dir, name, B):
    """a 2 % parameter step from the stored demos' optimum: the guard evaluates the previous solution, keeps the prediction (no status bit) and the solve is the unguarded
    one bit for bit - full and primal record, record and fp64 sensitivity inputs"""
    d, oc, mdl = setup(golden_dir, name)
    th = d["true_parameter"]
    T = d["control"].shape[1]
    x0 = np.repeat(d["state"][:1, 0], B, axis=0)
    x0[:, 1] += 0.01 * np.arange(B)
    sol = solve_at(mdl, x0, th, T)
    dem = (d["state"][:1].repeat(B, axis=0), d["control"][:1].repeat(B, axis=0))
    a = mdl.oc_pdp_grad(sol["control"], th, dem[0], dem[1], x=sol["state"], lam=sol["costate"], want_sens=True, want_riccati=True, want_predict_record=True)
    rng = np.random.default_rng(5)
    dth = th[None] * 0.02 * rng.uniform(-1, 1, (B, oc.p))
    th1 = th[None] + dth
    warm = (sol["state"], sol["control"], sol["costate"])
    for pred in (dict(record=a["predict_record"]), dict(record=a["predict_record"], primal=True), dict(dxdp=a["dxdp"], dudp=a["dudp"], riccati=a["riccati"])):
        on = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, guard=True, **pred), log_rows=6)
        off = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, guard=False, **pred), log_rows=6)
        assert bool(on["converged"].all()) and int((on["status"] & 512).sum()) == 0 and bool((on["status"] == off["status"]).all())
        for k in ("state", "control", "costate", "cost", "iterations", "log"):
            assert bool((on[k] == off[k]).all()), (sorted(pred), k)


@pytest.mark.parametrize("primal", [False, True])
def test_record_prediction_at_a_long_horizon(primal):
    """Rocket, T = 100 (C4's horizon): more than 64 nodes - the solver's residual pass runs in two lane groups that overlap by a node, the record prediction in several
    pool-sized batches, the prediction guard's pass over the previous solution on the evaluator while the runner forms the prediction (round 6).  The solve that applies the
    packed fp32 record inside the launch must start from the very point pdp_oc_predict_record_batched writes and therefore take the same iterations to the same
    optimum, with the guard (default) and without it; and it must agree with the solve from the plain warm start."""
    from pdp_amd import zoo, runtime as rt
    mdl = zoo.get("rocket", "irl")
    rng = np.random.default_rng(11)
    B, T = 6, 100
    th = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    x0 = np.zeros((B, 13))
    x0[:, 0:3] = np.array([10, -8, 5.0]) + 0.5 * rng.standard_normal((B, 3))
    x0[:, 3] = -0.5
    ang = 0.5 + 0.1 * rng.standard_normal(B)
    x0[:, 6], x0[:, 8], x0[:, 9] = np.cos(ang / 2), np.sin(ang / 2) / np.sqrt(2), -np.sin(ang / 2) / np.sqrt(2)
    sol = mdl.oc_solve_ms(x0, th, T)
    assert bool(sol["converged"].all())
    warm = (sol["state"], sol["control"], sol["costate"])
    want = "primal" if primal else True
    rec = mdl.oc_pdp_grad(sol["control"], th, sol["state"], sol["control"], x=sol["state"], lam=sol["costate"], want_predict_record=want)["predict_record"]
    th1 = th[None] * (1 + 0.01 * rng.uniform(-1, 1, (B, th.size)))
    dth = th1 - th[None]
    kw = dict(primal=True) if primal else {}
    xp, up, lp = mdl.oc_predict_from_record(sol["state"], sol["control"], sol["costate"], dth, rec, **kw)
    pre = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=(xp, up, lp))
    plain = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm)
    assert bool(pre["converged"].all()) and bool(plain["converged"].all())
    for guard in (True, False):
        inl = mdl.oc_solve_ms(x0, th1, T, tol=1e-10, warm=warm, predict=dict(dtheta=dth, record=rec, guard=guard, **kw))
        assert bool(inl["converged"].all())
        kept = (npy(inl["status"]) & 512) == 0                      # (a prediction the guard dropped starts from the previous solution: compared with the plain warm start)
        assert guard or bool(kept.all())
        for i in range(B):
            ref = pre if kept[i] else plain
            assert int(inl["iterations"][i]) == int(ref["iterations"][i]), (guard, i)
            for k in ("state", "control", "costate"):
                a, b_ = npy(inl[k][i]), npy(ref[k][i])
                assert np.abs(a - b_).max() <= 1e-12 * max(1.0, np.abs(b_).max()), (guard, i, k)
        for k in ("state", "control"):
            assert np.abs(npy(inl[k]) - npy(plain[k])).max() <= 1e-7 * max(1.0, np.abs(npy(plain[k])).max())
