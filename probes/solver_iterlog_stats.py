"""Would IPOPT's watchdog have anything to do on the hard cold solves (rocket, T = 100, zero guess: 383 / 512 converge within 300 iterations)?

IPOPT arms its watchdog after `watchdog_shortened_iter_trigger` = 10 consecutive iterations whose line search had to shorten the step (alpha < 1 here: no bounds).  The
kernel's iteration log (row: iteration, objective, inf_pr, inf_du, dw, alpha, grad(phi)'d, theta) of 256 such solves says how often that happens, and what else the
iterations are spent on (inertia corrections dw > 0, restorations alpha = 0).     python probes/solver_iterlog_stats.py  -> profiles/r06_solver_iterlog_stats.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from pdp_amd import JinEnv, zoo  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    rng.uniform(-0.5, 0.5, 256); rng.uniform(-0.45, 0.45, (256, 7))          # (the draws probes/solver_robustness.py makes before its rocket case)
    B, T, MI = 256, 100, 300
    x0 = np.zeros((512, 13))
    x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((512, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
    x0 = x0[:B]
    th = np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])
    mdl = zoo.get("rocket", "irl")
    s = mdl.oc_solve_ms(x0, th, T, tol=1e-8, max_iter=MI, log_rows=MI)
    log = s["log"].cpu().numpy()
    it = s["iterations"].cpu().numpy()
    conv = s["converged"].cpu().numpy().astype(bool)
    print("rocket T = %d, %d cold solves from the zero guess, tol 1e-8, max_iter %d: %d converged; iterations median %d" % (T, B, MI, conv.sum(), np.median(it)))
    short_frac, dw_frac, rest_frac, trig, trig_at, longest = [], [], [], 0, [], []
    for b in range(B):
        n = min(int(it[b]), MI)
        if n == 0:
            continue
        a, dw = log[b, :n, 5], log[b, :n, 4]
        shortened = (np.abs(a) < 1.0) & (a != 0.0)
        short_frac.append(shortened.mean()); dw_frac.append((dw > 0).mean()); rest_frac.append((a == 0.0).mean())
        run, best, first = 0, 0, None
        for k in range(n):
            run = run + 1 if shortened[k] else 0
            best = max(best, run)
            if run == 10 and first is None:
                first = k
        longest.append(best)
        if first is not None:
            trig += 1
            trig_at.append(first)
    short_frac, dw_frac, rest_frac, longest = map(np.array, (short_frac, dw_frac, rest_frac, longest))
    print("  share of a solve's iterations with a SHORTENED accepted step (0 < alpha < 1): mean %.3f, median %.3f, max %.3f" % (short_frac.mean(), np.median(short_frac), short_frac.max()))
    print("  share with an inertia correction (dw > 0): mean %.3f, median %.3f;   share ending in the restoration (alpha = 0): mean %.4f" % (dw_frac.mean(), np.median(dw_frac), rest_frac.mean()))
    print("  longest run of consecutive shortened iterations per solve: median %d, p95 %d, max %d" % (np.median(longest), np.percentile(longest, 95), longest.max()))
    print("  solves in which IPOPT's watchdog would have been armed (10 consecutive shortened iterations): %d of %d%s" %
          (trig, B, (" (first at iteration: median %d)" % np.median(trig_at)) if trig else ""))
    def short_alphas(bs):
        parts = []
        for b_ in bs:
            a_ = log[b_, :min(int(it[b_]), MI), 5]
            parts.append(np.abs(a_[(np.abs(a_) < 1.0) & (a_ != 0.0)]))
        v = np.concatenate(parts) if parts else np.array([])
        return float(np.median(v)) if v.size else float("nan")
    for name, m in (("converged", conv), ("at the iteration limit", ~conv)):
        if m.any():
            print("  %-24s %3d solves: shortened %.3f, dw > 0 %.3f, longest shortened run median %d; alpha of shortened steps median %.3g" %
                  (name, m.sum(), short_frac[m].mean(), dw_frac[m].mean(), np.median(longest[m]), short_alphas(np.where(m)[0])))
    b = int(np.where(~conv)[0][0]) if (~conv).any() else 0
    print("  one solve at the limit (sample %d), every 10th row: it, f, inf_pr, inf_du, dw, alpha" % b)
    for k in range(0, min(int(it[b]), MI), 10):
        r = log[b, k]
        print("     %4d  f %.6e  pr %.2e  du %.2e  dw %.1e  alpha %.3g" % (r[0], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
