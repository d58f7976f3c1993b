"""A/B timing of build variants of the multiple-shooting solver kernel (libraries under probes/_build/libms_var_<tag>.so, -DPDP_MS_TIMING): C3 warm solve."""
import sys, os, glob, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt
import bench
rng = np.random.default_rng(0)
B, T = 1024, 50
th_star = np.array(bench.THETA)
x0 = bench.synth_inputs(B, 5)[0]
theta1 = th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR)))
x0d, theta1 = rt.dev(x0), rt.dev(theta1)
ref = zoo.get("quadrotor", "irl")
demo = ref.oc_solve_ms(x0d, th_star, T)
warm = (demo["state"], demo["control"], demo["costate"])
COLS = ["first-chunk wait", "riccati", "chunk waits", "fwd steps", "dlam tail", "linesearch", "update", "total"]
for lib in sorted(glob.glob("probes/_build/libms_var_*.so")):
    mdl = rt.ModelLib(lib)
    sol = mdl.oc_solve_ms(x0d, theta1, T, warm=warm, log_rows=100)
    ms = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, theta1, T, warm=warm), reps=7, warm=2)
    cold = bench._event_ms(torch, lambda: mdl.oc_solve_ms(x0d, th_star, T), reps=3, warm=1)
    it = sol["iterations"].cpu().numpy(); log = sol["log"].cpu().numpy()
    per = np.array([log[b, :it[b], :].mean(axis=0) for b in range(B) if it[b] > 0])
    print("%s: warm %.3f ms cold %.3f ms | " % (os.path.basename(lib), ms, cold) + " | ".join("%s %d" % (c, v) for c, v in zip(COLS, np.median(per, axis=0))), flush=True)
