"""EXPERIMENT (CPU, oracle only): what would IPOPT's watchdog make of the cold rocket solves at T = 100 that crawl?   profiles/r06_watchdog_experiment.txt

probes/solver_iterlog_stats.py: of 256 such solves 67 sit at the 300-iteration limit, 93 % of their iterations accepted steps of alpha ~ 1e-3, and IPOPT's trigger (10
consecutive shortened iterations) is met in 199 of 256.  The kernels do not have the watchdog; oracle/ipopt_ms.solve(watchdog=True) restates it from memory of IPOPT's
BacktrackingLineSearch (no source, no IPOPT to pin it on: an experiment, labelled so).  Same problems as probes/solver_robustness.py's C4 case, first N of them, each solved
without and with it (max_iter 600, tol 1e-8): iterations, final cost, watchdog procedures started / successful.      python probes/watchdog_experiment.py [N]"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

sys.path.insert(0, os.getcwd())


def problem_set():
    from oracle import models
    rng = np.random.default_rng(0)
    rng.uniform(-0.5, 0.5, 256); rng.uniform(-0.45, 0.45, (256, 7))          # (the draws probes/solver_robustness.py makes before its rocket case)
    x0 = np.zeros((512, 13))
    x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((512, 3))
    x0[:, 3] = -0.1
    x0[:, 6:10] = models.to_quaternion(1.5, [0, 0, 1])
    return x0, np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0])


def one(args):
    b, wd, soc = args
    from oracle import models, pdp_oracle as po, ipopt_ms
    st = models.IRL_SETUP["rocket"]
    oc = po.make_oc(models.REGISTRY["rocket"](**st["kwargs"]), st["dt"])
    x0, th = problem_set()
    log = []
    t0 = time.time()
    try:
        r = ipopt_ms.solve(oc, x0[b], 100, th, tol=1e-8, max_iter=600, log=log, watchdog=wd, soc=soc)
        out = (b, wd, soc, True, r["iterations"], r["cost"], r["watchdog_starts"], r["watchdog_successes"], r["restorations"])
    except RuntimeError as ex:
        tags = [l.get("wd") for l in log]
        out = (b, wd, soc, False, len(log), log[-1]["f"] if log else float("nan"), tags.count("start") + tags.count("success") - sum(1 for i, t in enumerate(tags) if t == "success" and i and tags[i - 1] in ("start", "trial")),
               tags.count("success"), sum(1 for l in log if l.get("restoration")))
    return out + (time.time() - t0,)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    jobs = [(b, wd, soc) for b in range(N) for wd, soc in ((False, False), (True, False), (False, True), (True, True))]
    with Pool(min(8, os.cpu_count() or 1)) as p:
        res = p.map(one, jobs, chunksize=1)
    by = {}
    for r in res:
        by.setdefault((r[1], r[2]), []).append(r)
    print("rocket T = 100, zero guess, tol 1e-8, max_iter 600, the first %d problems of probes/solver_robustness.py's C4 case; oracle/ipopt_ms.py on the CPU" % N)
    base = {r[0]: r for r in by[(False, False)]}
    for key, name in (((False, False), "as the kernels iterate (no watchdog, no SOC)"), ((True, False), "with the watchdog"), ((False, True), "with the second-order correction"),
                      ((True, True), "with both (IPOPT's defaults)")):
        rows = sorted(by[key])
        it = np.array([r[4] for r in rows], dtype=float)
        conv = np.array([r[3] for r in rows])
        same = sum(1 for r in rows if r[3] and base[r[0]][3] and abs(r[5] - base[r[0]][5]) <= 1e-6 * max(1.0, abs(base[r[0]][5])))
        both = sum(1 for r in rows if r[3] and base[r[0]][3])
        print("  %-46s converged %2d / %2d within 600 (%2d within 300)   iterations median %3.0f  mean %5.1f  max %3.0f   watchdog procedures %3d started, %3d successful   "
              "same optimum as the first row on %d of the %d both solve" %
              (name, conv.sum(), N, int(((it <= 300) & conv).sum()), np.median(it), it.mean(), it.max(), sum(r[6] for r in rows), sum(r[7] for r in rows), same, both))
    print("  per problem: iterations (c = converged)  no watchdog | watchdog | SOC | both      cost without / with watchdog")
    for b in range(N):
        rs = [next(r for r in by[k] if r[0] == b) for k in ((False, False), (True, False), (False, True), (True, True))]
        print("   %3d   %s   %.6f / %.6f" % (b, " | ".join("%3d%s" % (r[4], "c" if r[3] else " ") for r in rs), rs[0][5], rs[1][5]))


if __name__ == "__main__":
    main()
