"""Class surface (ocsolver.solve_batch) on the cold rocket solves at T = 100 with and without PDP_MS_WITH_WATCHDOG: how many rows the kernel solves, how many the single-shooting
fallback, wall clock (appended to profiles/r06_solver_robustness.txt).   python probes/class_surface_watchdog.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch
sys.path.insert(0, os.path.join(os.getcwd(), 'probes'))
import solver_robustness as sr
from pdp_amd import ocsolver
for name, system, x0, th, T in sr.cases():
    if system != 'rocket': continue
    oc = sr.make_oc(system)
    for wd in (False, True):
        ocsolver.solve_batch(oc, x0[:8], T, th, tol=1e-9, watchdog=wd)
        torch.cuda.synchronize(); t=time.perf_counter()
        sol = ocsolver.solve_batch(oc, x0, T, th, tol=1e-9, watchdog=wd)
        torch.cuda.synchronize(); dt=time.perf_counter()-t
        conv = sol['converged'].cpu().numpy().astype(bool); by = sol['method_ms'].cpu().numpy().astype(bool)
        print('%s class surface watchdog=%s: %d / %d converged (%d by the kernel, %d through the single-shooting fallback) in %.1f ms wall clock; cost median %.4f' % (name, wd, conv.sum(), len(conv), by.sum(), (conv & ~by).sum(), dt*1e3, float(np.median(sol['cost'].cpu().numpy()))))
