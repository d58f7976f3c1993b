"""Cycle stamps of the runner / evaluator kernel (oc_pdp_fused3_kernel, -DPDP_PHASE_TIMING build), trajectory 0."""
import sys, os, subprocess, numpy as np
os.environ["PDP_FUSED_VARIANT"] = "3"
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import codegen, zoo, runtime
import bench
SYSTEM = os.environ.get('PDP_SYSTEM', 'quadrotor')      # round 6: PDP_SYSTEM=rocket PDP_B=512 -> one GPU's shard of C4 (rocket T = 100, TPW = 2)
pb = zoo.make_problem(SYSTEM, 'irl'); _, info = codegen.write_header(pb)
out = '/tmp/libtiming3.so'
EXTRA = [a for a in os.environ.get('PDP_EXTRA', '').split() if a]
subprocess.run([codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + EXTRA + ['-DPDP_PHASE_TIMING', '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'], '-I', codegen.CSRC,
                os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
mdl = runtime.ModelLib(out)
for B in (int(os.environ.get('PDP_B', '1024')),):
    if SYSTEM == 'rocket':
        from pdp_amd import JinEnv
        rng = np.random.default_rng(0)
        T = 100
        x0 = np.zeros((B, 13)); x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3)); x0[:, 3] = -0.1; x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
        u = np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3))
        x0, u, dx, du = (torch.as_tensor(a, device='cuda') for a in (x0, u, np.zeros((B, T + 1, 13)), np.zeros((B, T, 3))))
        th = torch.tensor([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0], dtype=torch.float64, device='cuda')
    else:
        x0, u, dx, du = (torch.as_tensor(a, device='cuda') for a in bench.synth_inputs(B, 1000))
        th = torch.tensor(bench.THETA, dtype=torch.float64, device='cuda')
    big = torch.zeros(B + 64 + 4 * B, dtype=torch.float64, device='cuda')
    bufs = {'loss': big[:B]}
    for _ in range(3):
        o = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs)
    torch.cuda.synchronize()
    if os.environ.get('PDP_GIVEN'):      # trajectory and costates handed in: no rollout on the runner, no costate chain on the evaluator
        xg, lg = o['x'].clone(), o['lam'].clone()
        for _ in range(3):
            o = mdl.oc_pdp_grad(u, th, dx, du, x=xg, lam=lg, buffers={'loss': big[:B]})
        torch.cuda.synchronize()
    st = big[B:].view(torch.int64).cpu().numpy()
    r, e = st[:13], st[16:29]
    t0 = r[0]
    print('B=%d runner   : rollout %d | terminal (wait) %d | backward %d | forward %d | tail %d | total %d | waiting for the evaluator %d' %
          (B, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[5] - r[0], r[12]))
    print('       evaluator: start +%d | wait for trajectory until +%d | terminal done +%d | backward chunks done +%d | forward chunks done +%d | waiting %d' %
          (e[0] - t0, e[1] - t0, e[2] - t0, e[3] - t0, e[4] - t0, e[12]))
    f = st[32:48]
    if 'PDP_PHASE_TIMING_FINE' in os.environ.get('PDP_EXTRA', ''): print('       last backward step (t = 0): entry->T0 %d | PF,PY2 issued %d | FY,Q2 issued %d | Qux,Pn issued %d | Quu in LDS %d | Z ready %d | K,IK,P- issued %d | P in LDS %d | symmetrised %d | stores %d || previous step exit -> this exit %d' % (f[0]-f[9], f[1]-f[0], f[2]-f[1], f[3]-f[2], f[4]-f[3], f[5]-f[4], f[6]-f[5], f[7]-f[6], f[8]-f[7], f[10]-f[8], f[10]-f[11]))
    per = st[64:64 + 4 * B].reshape(B, 4)
    t00 = per[:, 0].min()
    print('       all trajectories (100 MHz clock -> us): first start 0, last start %.2f us | first end %.2f, last end %.2f us | cycles min %d median %d max %d | wait median %d max %d' %
          ((per[:, 0].max() - t00) / 100.0, (per[:, 1].min() - t00) / 100.0, (per[:, 1].max() - t00) / 100.0, per[:, 2].min(), np.median(per[:, 2]), per[:, 2].max(),
           np.median(per[:, 3]), per[:, 3].max()))
    order = np.argsort(per[:, 1])
    print('       slowest five trajectories:', [(int(i), int(per[i, 2]), round((per[i, 0] - t00) / 100.0, 2)) for i in order[-5:]])
