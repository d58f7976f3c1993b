import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from pdp_amd import ocsolver
from test_gpu_ocsolver import make_oc
oc = make_oc("cartpole")
rng = np.random.default_rng(0)
B, T = 256, 50
th_star = np.array([0.5, 0.5, 1, 1, 6, 1, 1.0])
x0 = np.zeros((B, 4)); x0[:, 1] = rng.uniform(-0.5, 0.5, B)
demo = ocsolver.solve_batch(oc, x0, T, th_star, want_gains=True)
print('demo conv', int(demo['converged'].sum()))
theta = th_star[None, :] + rng.uniform(-0.1, 0.1, (B, 7))
sol = ocsolver.solve_batch(oc, x0, T, theta, warm_start=demo, want_gains=True, neighbor_retries=0)
conv = sol['converged'].cpu().numpy(); bad=np.where(~conv)[0]
print('conv', conv.sum(), 'iters', sol['iterations'], 'bad gnorm', sol['grad_norm'].cpu().numpy()[bad], 'cost', sol['cost'].cpu().numpy()[bad], 'demo cost', demo['cost'].cpu().numpy()[bad])
i=bad[0]
s = ocsolver._solve(oc, x0[i:i+1], T, theta[i], u_init=demo['control'][i:i+1], force_init=True, print_level=1, max_iter=30)
