"""Solver robustness as data (round-5 verdict, item 6): cold solves of `OCSys.ocSolver` (pdp_oc_solve_ms_batched, the reference's multiple-shooting NLP from IPOPT's zero
guess, PDP/PDP.py:121-220) at the BASELINE batch scales - how many converge, in how many Newton iterations, with which status bits - and what becomes of the rest when
they take the class surface's route (ocsolver.solve_batch: restoration inside the kernel, then single shooting for what is still unconverged).

    C2  cart-pole   T = 50   B = 256    per-sample theta = theta* +- U(-0.45, 0.45)           
    C3  quadrotor   T = 50   B = 1024   bench.synth_inputs initial poses
    C4  rocket      T = 100  B = 512 and 4096   powered-landing initial states of bench.py's C4
Both at the kernel's parity tolerance 1e-10 and at IPOPT's default 1e-8 (PDP.py:178-182 passes no tolerance), with and without the second-order correction.
Prints a table and writes JSON (argv[1], default gpurun_out/solver_robustness.json)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402

import bench  # noqa: E402
from pdp_amd import JinEnv, PDP, ocsolver, zoo  # noqa: E402

BITS = {1: "NONFINITE", 2: "PIVOT", 4: "RESTORATION (left unrestored)", 8: "MAXITER", 16: "INERTIA", 64: "INTERNAL", 128: "RESTORED (informational)", 1024: "SOC step taken (informational)"}


def make_oc(system):
    env, dt = zoo.make_env(system, "irl")
    oc = PDP.OCSys(system)
    from pdp_amd import sx
    oc.setAuxvarVariable(sx.vertcat(env.dyn_auxvar, env.cost_auxvar))
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U)
    oc.setDyn(env.X + dt * env.f)
    oc.setPathCost(env.path_cost)
    oc.setFinalCost(env.final_cost)
    return oc


def cases():
    rng = np.random.default_rng(0)
    B = 256
    x0 = np.zeros((B, 4))
    x0[:, 1] = rng.uniform(-0.5, 0.5, B)
    th = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])[None] + rng.uniform(-0.45, 0.45, (B, 7))
    yield "C2_cartpole_T50_B256", "cartpole", x0, th, 50
    yield "C3_quadrotor_T50_B1024", "quadrotor", bench.synth_inputs(1024, 5)[0], np.array(bench.THETA), 50
    for B in (512, 4096):
        x0 = np.zeros((B, 13))
        x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3))
        x0[:, 3] = -0.1
        x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        yield "C4_rocket_T100_B%d" % B, "rocket", x0, np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]), 100


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/solver_robustness.json"
    res = {}
    for name, system, x0, th, T in cases():
        mdl = zoo.get(system, "irl")
        B = x0.shape[0]
        entry = {"batch": B, "horizon": T, "kernel": {}}
        for tol in (1e-10, 1e-8):
            for soc in (False, True):
                key = "tol_%g%s" % (tol, "_soc" if soc else "")
                mdl.oc_solve_ms(x0, th, T, tol=tol, soc=soc)
                ms = float(bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0, th, T, tol=tol, soc=soc), reps=3, warm=0))
                s = mdl.oc_solve_ms(x0, th, T, tol=tol, soc=soc)
                conv = s["converged"].cpu().numpy().astype(bool)
                it = s["iterations"].cpu().numpy()
                st = s["status"].cpu().numpy()
                entry["kernel"][key] = {"converged": int(conv.sum()), "of": B, "rate": float(conv.mean()), "ms": ms,
                                        "iterations": {"mean": float(it.mean()), "median": float(np.median(it)), "p95": float(np.percentile(it, 95)), "max": int(it.max())},
                                        "iterations_of_converged": {"mean": float(it[conv].mean()) if conv.any() else None, "max": int(it[conv].max()) if conv.any() else None},
                                        "status_bits": {BITS.get(b, str(b)): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 64, 128, 1024) if ((st & b) != 0).any()},
                                        "cost_of_converged_min_median_max": [float(v) for v in np.percentile(s["cost"].cpu().numpy()[conv], [0, 50, 100])] if conv.any() else None}
                print("%-24s %-14s converged %4d / %4d  iterations mean %.1f median %.0f p95 %.0f max %d  %.3f ms  status %s" %
                      (name, key, conv.sum(), B, it.mean(), np.median(it), np.percentile(it, 95), it.max(), ms, entry["kernel"][key]["status_bits"]))
        if system == "rocket":
            # IPOPT's own iteration limit is 3000 (the reference passes none, PDP.py:178-182); this solver's default is 300.  What the longer leash buys:
            for mi in (1000, 3000):
                s = mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=mi)
                conv = s["converged"].cpu().numpy().astype(bool)
                it = s["iterations"].cpu().numpy()
                st = s["status"].cpu().numpy()
                ms = float(bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=mi), reps=1, warm=0))
                entry["kernel"]["tol_1e-08_max_iter_%d" % mi] = {"converged": int(conv.sum()), "of": B, "rate": float(conv.mean()), "ms": ms,
                                                                "iterations": {"mean": float(it.mean()), "median": float(np.median(it)), "p95": float(np.percentile(it, 95)), "max": int(it.max())},
                                                                "status_bits": {BITS.get(b, str(b)): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 64, 128, 1024) if ((st & b) != 0).any()}}
                print("%-24s max_iter %-5d converged %4d / %4d  iterations mean %.1f median %.0f p95 %.0f max %d  %.1f ms  status %s" %
                      (name, mi, conv.sum(), B, it.mean(), np.median(it), np.percentile(it, 95), it.max(), ms, entry["kernel"]["tol_1e-08_max_iter_%d" % mi]["status_bits"]))
        # PDP_MS_WITH_WATCHDOG (round 6, third session): IPOPT's watchdog in the kernel's line search, at this solver's limit and at the longer leashes
        for mi in ((300, 1000, 3000) if system == "rocket" else (300,)):
            s = mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=mi, watchdog=True)
            conv = s["converged"].cpu().numpy().astype(bool)
            it = s["iterations"].cpu().numpy()
            st = s["status"].cpu().numpy()
            ms = float(bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=mi, watchdog=True), reps=1, warm=0))
            key = "tol_1e-08_max_iter_%d_watchdog" % mi
            entry["kernel"][key] = {"converged": int(conv.sum()), "of": B, "rate": float(conv.mean()), "ms": ms,
                                    "iterations": {"mean": float(it.mean()), "median": float(np.median(it)), "p95": float(np.percentile(it, 95)), "max": int(it.max())},
                                    "watchdog_procedures_started_in": int(((st & 2048) != 0).sum()),
                                    "status_bits": {BITS.get(b, str(b)): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 64, 128, 1024) if ((st & b) != 0).any()}}
            print("%-24s max_iter %-5d WATCHDOG converged %4d / %4d  iterations mean %.1f median %.0f p95 %.0f max %d  %.1f ms  started in %d  status %s" %
                  (name, mi, conv.sum(), B, it.mean(), np.median(it), np.percentile(it, 95), it.max(), ms, entry["kernel"][key]["watchdog_procedures_started_in"], entry["kernel"][key]["status_bits"]))
        if system == "rocket":              # both switches: IPOPT's default pair, as restated
            for mi in (300, 1000):
                s = mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=mi, watchdog=True, soc=True)
                conv = s["converged"].cpu().numpy().astype(bool)
                it = s["iterations"].cpu().numpy()
                st = s["status"].cpu().numpy()
                key = "tol_1e-08_max_iter_%d_watchdog_soc" % mi
                entry["kernel"][key] = {"converged": int(conv.sum()), "of": B, "rate": float(conv.mean()),
                                        "iterations": {"mean": float(it.mean()), "median": float(np.median(it)), "p95": float(np.percentile(it, 95)), "max": int(it.max())},
                                        "status_bits": {BITS.get(b, str(b)): int(((st & b) != 0).sum()) for b in (1, 2, 4, 8, 16, 64, 128, 1024) if ((st & b) != 0).any()}}
                print("%-24s max_iter %-5d WATCHDOG + SOC converged %4d / %4d  iterations mean %.1f median %.0f p95 %.0f max %d  status %s" %
                      (name, mi, conv.sum(), B, it.mean(), np.median(it), np.percentile(it, 95), it.max(), entry["kernel"][key]["status_bits"]))
        # the class surface's route (what a user of OCSys.ocSolver_batch gets): kernel, then single shooting for the rows it left unconverged
        oc = make_oc(system)
        sol = ocsolver.solve_batch(oc, x0, T, th, tol=1e-9)
        conv = sol["converged"].cpu().numpy().astype(bool)
        by_ms = sol["method_ms"].cpu().numpy().astype(bool)
        entry["class_surface_solve_batch_tol_1e-9"] = {"converged": int(conv.sum()), "of": B, "converged_by_the_multiple_shooting_kernel": int(by_ms.sum()),
                                                       "came_home_through_the_single_shooting_fallback": int((conv & ~by_ms).sum()), "still_unconverged": int((~conv).sum())}
        print("%-24s class surface: %d / %d converged (%d by the kernel, %d through the fallback, %d not)" %
              (name, conv.sum(), B, by_ms.sum(), (conv & ~by_ms).sum(), (~conv).sum()))
        res[name] = entry
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
