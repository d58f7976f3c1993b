import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from pdp_amd import ocsolver
from test_gpu_ocsolver import make_oc
G='tests/golden/'
for name, j in (('cartpole', 0), ('robotarm', 0)):
    d=np.load(G+'demos_%s.npz'%name); tr=np.load(G+'irltrace_%s.npz'%name)
    oc=make_oc(name); th=tr['param'][j]
    print(name, 'theta', th)
    sol = ocsolver.solve_batch(oc, d['state'][:,0], d['control'].shape[1], th, u_init=d['control'], print_level=0)
    print('iters', sol['iterations'], 'conv', sol['converged'].cpu().numpy(), 'gn', sol['grad_norm'].cpu().numpy(), 'cost', sol['cost'].cpu().numpy())
