import sys, os, numpy as np, time
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import PDP, JinEnv, zoo, ocsolver
from pdp_amd.sx import vertcat
G='tests/golden/'
for name in ['pendulum','cartpole','robotarm','quadrotor','rocket']:
    env, dt = zoo.make_env(name, 'irl')
    oc = PDP.OCSys(name)
    oc.setAuxvarVariable(vertcat(env.dyn_auxvar, env.cost_auxvar)); oc.setStateVariable(env.X); oc.setControlVariable(env.U)
    oc.setDyn(env.X + dt*env.f); oc.setPathCost(env.path_cost); oc.setFinalCost(env.final_cost)
    d=np.load(G+'demos_%s.npz'%name)
    t0=time.time()
    sol = ocsolver.solve_batch(oc, d['state'][:,0], d['control'].shape[1], d['true_parameter'], print_level=int(os.environ.get('PL','0')))
    torch.cuda.synchronize(); dt_=time.time()-t0
    x,u,l = [sol[k].cpu().numpy() for k in ('state','control','costate')]
    print(name,'iters',sol['iterations'],'conv',sol['converged'].cpu().numpy(),'gnorm %.2e'%float(sol['grad_norm'].max()), 'cost', sol['cost'].cpu().numpy()[:2], d['cost'][:2],
          'err x %.2e u %.2e lam %.2e'%(np.abs(x-d['state']).max(), np.abs(u-d['control']).max(), np.abs(l-d['costate']).max()/np.abs(d['costate']).max()), '%.2fs'%dt_)
