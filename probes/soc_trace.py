"""Second-order correction: the solver kernel's iteration log beside the restatement's (oracle/ipopt_ms.py: solve(soc=True)) on stored demos - where they part, if they do.
    python probes/soc_trace.py [system demo]..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ipopt_ms, models, pdp_oracle as po      # noqa: E402
from pdp_amd import zoo                                    # noqa: E402

cases = [("cartpole", 0), ("robotarm", 0), ("robotarm", 1), ("quadrotor", 0), ("rocket", 0), ("robotarm", 3)]
if len(sys.argv) > 2:
    cases = [(sys.argv[i], int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
for name, demo in cases:
    d = np.load(os.path.join(ROOT, "tests", "golden", "demos_%s.npz" % name))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    T = d["control"].shape[1]
    log = []
    ref = ipopt_ms.solve(oc, d["state"][demo, 0], T, d["true_parameter"], log=log, soc=True)
    sol = zoo.get(name, "irl").oc_solve_ms(d["state"][demo:demo + 1, 0], d["true_parameter"], T, tol=1e-10, log_rows=len(log) + 8, soc=True)
    kl = sol["log"][0].cpu().numpy()
    print("== %s demo %d: oracle %d iterations (%d corrections tried), kernel %d, status %d" % (name, demo, ref["iterations"], ref["soc_steps"], int(sol["iterations"][0]), int(sol["status"][0])))
    shown = 0
    for r, l in zip(kl, log):
        a = -l["alpha"] if l["soc_taken"] else l["alpha"]
        bad = r[5] != a or abs(r[1] - l["f"]) > 1e-9 * max(1.0, abs(l["f"])) or abs(r[7] - l["theta"]) > 1e-9 * max(1.0, l["theta"])
        if bad or shown:
            print("  it %3d  oracle f %.10e th %.6e dw %.2e alpha %+.4e soc %d | kernel f %.10e th %.6e dw %.2e alpha %+.4e gd %.6e (oracle gd %.6e)" % (
                l["it"], l["f"], l["theta"], l["dw"], a, l["soc"], r[1], r[7], r[4], r[5], r[6], l["gd"]))
            shown += 1
            if shown > 3:
                break
    if not shown:
        print("  identical rows")
