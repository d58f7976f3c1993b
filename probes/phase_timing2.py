"""Cycle stamps of the two-wave fused kernel (oc_pdp_fused2_kernel, -DPDP_PHASE_TIMING build), workgroup 0, B = 512 and 1024."""
import sys, os, subprocess, numpy as np
os.environ["PDP_FUSED_VARIANT"] = "2"
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import codegen, zoo, runtime
import bench
pb = zoo.make_problem('quadrotor', 'irl'); _, info = codegen.write_header(pb)
out = '/tmp/libtiming2.so'
subprocess.run([codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + ['-DPDP_PHASE_TIMING', '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'], '-I', codegen.CSRC,
                os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
mdl = runtime.ModelLib(out)
for B in (512, 1024):
    x0, u, dx, du = (torch.as_tensor(a, device='cuda') for a in bench.synth_inputs(B, 1000))
    th = torch.tensor(bench.THETA, dtype=torch.float64, device='cuda')
    big = torch.zeros(B + 64, dtype=torch.float64, device='cuda')
    bufs = {'loss': big[:B]}
    for _ in range(3):
        o = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs)
    torch.cuda.synchronize()
    st = big[B:].view(torch.int64).cpu().numpy()
    s = st[:6]
    print('B=%d wave A stamps: rollout %d | terminal %d | backward %d | forward %d | total %d' % (B, s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[4] - s[0]))
    a = st[16:20]
    print('   backward totals: evalA+costates %d | evalB (wave B; A waits) %d | Riccati loops %d' % (a[0], a[1], a[2]))
    fa, fb = st[8:16], st[32:40]
    for nm, f in (('A', fa), ('B', fb)):
        if nm == 'B': print('   wave B part 1: MFMAs issued at +%d, m x m solve %d, rest %d' % (f[6] - f[0], f[7] - f[6], f[1] - f[7]))
        print('   step t=20 wave %s: part1 %d | wait1 %d | part2 %d | wait2 %d | symmetrise %d | step total %d' % (nm, f[1] - f[0], f[2] - f[1], f[3] - f[2], f[4] - f[3], f[5] - f[4], f[5] - f[0]))
