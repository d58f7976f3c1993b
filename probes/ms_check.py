"""GPU diagnostic: the multiple-shooting OC solver kernel (pdp_oc_solve_ms_batched) against its CPU restatement oracle/ipopt_ms.py,
iteration by iteration, on the reference's stored demos; then timings on the BASELINE shapes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import models, pdp_oracle as po, ipopt_ms
from pdp_amd import zoo

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in sys.argv[1:] or ["pendulum", "rocket", "quadrotor", "robotarm", "cartpole"]:
    d = np.load(os.path.join(G, "demos_%s.npz" % name))
    mdl = zoo.get(name, "irl")
    T = d["control"].shape[1]
    t0 = time.perf_counter()
    sol = mdl.oc_solve_ms(d["state"][:, 0], d["true_parameter"], T, log_rows=80)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    x, u, lam = (sol[k].cpu().numpy() for k in ("state", "control", "costate"))
    print("%-9s conv %s iters %s status %s  %.2f ms" % (name, sol["converged"].cpu().numpy().astype(int), sol["iterations"].cpu().numpy(), sol["status"].cpu().numpy(), dt * 1e3))
    print("          dcost %.2e dx %.2e du %.2e dlam %.2e" % (np.abs(sol["cost"].cpu().numpy() - d["cost"]).max(), np.abs(x - d["state"]).max(),
                                                               np.abs(u - d["control"]).max(), np.abs(lam - d["costate"]).max()))
    st = models.IRL_SETUP[name]
    oc = po.make_oc(models.REGISTRY[name](**st["kwargs"]), st["dt"])
    log = []
    try:
        ref = ipopt_ms.solve(oc, d["state"][0, 0], T, d["true_parameter"], log=log)
        print("          oracle: iters %d; kernel vs oracle dx %.2e dlam %.2e" % (ref["iterations"], np.abs(x[0] - ref["state_traj_opt"]).max(), np.abs(lam[0] - ref["costate_traj_opt"]).max()))
    except RuntimeError as ex:
        print("          oracle:", ex)
    kl = sol["log"][0].cpu().numpy()
    bad = False
    for r, l in zip(kl, log):
        rel = lambda a, b: abs(a - b) / max(1e-300, abs(a), abs(b))
        if r[5] != l["alpha"] or rel(r[4], l["dw"]) > 1e-12 or rel(r[1], l["f"]) > 1e-7:
            bad = True
    if bad or len(log) != int(sol["iterations"][0]):
        print("          iteration log differs (kernel | oracle):")
        for i in range(max(len(log), int(sol["iterations"][0]) + 1)):
            r = kl[i] if i < len(kl) else np.zeros(8)
            l = log[i] if i < len(log) else None
            print("   k it %2d f %.10e pr %.2e du %.2e dw %.2e a %.4g gd %.6e th %.6e" % tuple(r), "| o f %.10e pr %.2e du %.2e dw %.2e a %.4g gd %.6e th %.6e" %
                  (l["f"], l["inf_pr"], l["inf_du"], l["dw"], l["alpha"], l["gd"], l["theta"]) if l else "")

# timings: C2 (cart-pole, 256 per-sample theta, cold and warm), C3-sized quadrotor batch
rng = np.random.default_rng(0)
for name, B in (("cartpole", 256), ("quadrotor", 1024), ("rocket", 512)):
    d = np.load(os.path.join(G, "demos_%s.npz" % name))
    mdl = zoo.get(name, "irl")
    T = 50 if name != "rocket" else 100
    x0 = np.repeat(d["state"][:1, 0], B, axis=0) * (1 + 0.05 * rng.standard_normal((B, mdl.n)))
    th = d["true_parameter"][None] * (1 + 0.05 * rng.uniform(-1, 1, (B, mdl.p)))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sol = mdl.oc_solve_ms(x0, th, T)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = sol["iterations"].cpu().numpy()
    print("%-9s B=%d T=%d cold: %.2f ms, converged %d/%d, iterations min/mean/max %d/%.1f/%d, status %s" % (name, B, T, dt * 1e3, int(sol["converged"].sum()), B, it.min(), it.mean(), it.max(),
          np.unique(sol["status"].cpu().numpy())))
    th2 = th * (1 + 0.02 * rng.uniform(-1, 1, th.shape))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s2 = mdl.oc_solve_ms(x0, th2, T, warm=(sol["state"], sol["control"], sol["costate"]))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    it = s2["iterations"].cpu().numpy()
    print("%-9s B=%d T=%d warm (theta perturbed 2%%): %.2f ms, converged %d/%d, iterations min/mean/max %d/%.1f/%d" % (name, B, T, dt * 1e3, int(s2["converged"].sum()), B, it.min(), it.mean(), it.max()))
