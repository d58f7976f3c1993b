"""Cycle breakdown of the multiple-shooting OC solver kernel per iteration (-DPDP_MS_TIMING build: the iteration log carries cycles per phase).
Workloads: bench.py's C2 / C3 IRL iterations (warm solve from the solution at theta*, cold solve).  PDP_EXTRA adds compile flags."""
import sys, os, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import codegen, zoo, runtime as rt
import bench
EXTRA = [a for a in os.environ.get('PDP_EXTRA', '').split() if a]
VARIANT = os.environ.get("PDP_MS_VARIANT", "2")
COLS = (["residuals", "bwd misc", "bwd eval", "riccati", "fwd eval", "fwd steps", "linesearch", "total"] if VARIANT == "1" else
        ["first-chunk wait", "riccati", "chunk waits", "fwd steps", "dlam tail", "linesearch+update", "sweep prologues", "total"])
rng = np.random.default_rng(0)
for system, B, T in (("cartpole", 256, 50), ("quadrotor", 1024, 50), ("quadrotor", 256, 50)):
    pb = zoo.make_problem(system, 'irl'); _, info = codegen.write_header(pb)
    os.makedirs('probes/_build', exist_ok=True)
    out = 'probes/_build/libms_timing_%s.so' % system
    flags = codegen.HIP_FLAGS
    if not (os.path.exists(out) and os.environ.get('PDP_PREBUILT')):
      subprocess.run([codegen.HIPCC] + flags + codegen.OC_EXTRA_FLAGS + EXTRA + ['-DPDP_MS_TIMING', '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'], '-I', codegen.CSRC,
                    os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
    mdl = rt.ModelLib(out)
    ref = zoo.get(system, "irl")
    if system == "cartpole":
        th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
        x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-0.5, 0.5, B)
        theta1 = th_star[None] + rng.uniform(-0.05, 0.05, (B, 7))
    else:
        th_star = np.array(bench.THETA)
        x0 = bench.synth_inputs(B, 5)[0]
        theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR)))
    x0d, theta1 = rt.dev(x0), rt.dev(theta1)
    demo = ref.oc_solve_ms(x0d, th_star, T)
    warm = (demo["state"], demo["control"], demo["costate"])
    for label, kw, th in (("warm", dict(warm=warm), theta1),) + ((("cold", {}, th_star),) if B != 256 or system == "cartpole" else ()):
        for _ in range(2):
            sol = mdl.oc_solve_ms(x0d, th, T, log_rows=100, **kw)
        ms = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, th, T, log_rows=100, **kw), reps=5, warm=1)
        prod = bench._event_ms(torch, lambda: ref.oc_solve_ms(x0d, th, T, **kw), reps=5, warm=1)
        it = sol["iterations"].cpu().numpy()
        log = sol["log"].cpu().numpy()
        print("%s %s B=%d: timing build %.3f ms, product build %.3f ms; iterations mean %.2f max %d; converged %d" % (system, label, B, ms, prod, it.mean(), it.max(), int(sol["converged"].sum())))
        slow = int(np.argmax(it))
        for b in (0, slow):
            print("  trajectory %d (%d iterations): cycles per iteration" % (b, it[b]))
            for k in range(min(int(it[b]), 6)):
                print("    it %d: " % k + " | ".join("%s %d" % (c, v) for c, v in zip(COLS, log[b, k])))
        n_it = np.minimum(it, 100)
        tot = np.array([log[b, :n_it[b], :].sum(axis=0) for b in range(B)])
        print("  sum over iterations, median over the batch: " + " | ".join("%s %d" % (c, v) for c, v in zip(COLS, np.median(tot, axis=0))))
        per = np.array([log[b, :n_it[b], :].mean(axis=0) for b in range(B) if n_it[b] > 0])
        print("  per iteration, median over the batch:      " + " | ".join("%s %d" % (c, v) for c, v in zip(COLS, np.median(per, axis=0))))
        if VARIANT != "1":      # the evaluator's cumulative cycle counters sit in the second half of the log
            ECOLS = ["trial passes", "updates", "bwd chunk evals", "fwd chunk evals", "dlam", "terminal", "waits in sweeps", "n trials"]
            ev = np.array([log[b, 50 + max(n_it[b], 1) - 1, :] / max(n_it[b], 1) for b in range(B) if 0 < n_it[b] <= 50])
            if len(ev):
                print("  evaluator, per iteration, median:          " + " | ".join("%s %d" % (c, v) for c, v in zip(ECOLS, np.median(ev, axis=0))))
