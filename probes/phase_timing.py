import sys, os, subprocess, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pdp_amd import codegen, zoo, runtime
import bench
pb = zoo.make_problem('quadrotor','irl'); _, info = codegen.write_header(pb)
out = '/tmp/libtiming.so'
EXTRA=[a for a in os.environ.get('PDP_EXTRA','').split() if a]
subprocess.run([codegen.HIPCC]+codegen.HIP_FLAGS+codegen.OC_EXTRA_FLAGS+EXTRA+['-DPDP_PHASE_TIMING','-DPDP_MODEL_HEADER="generated/%s.h"'%info['name'],'-I',codegen.CSRC,os.path.join(codegen.CSRC,'pdp_model.hip'),'-o',out],check=True)
mdl = runtime.ModelLib(out)
B=1024
x0,u,dx,du = (torch.as_tensor(a,device='cuda') for a in bench.synth_inputs(B,1000))
th = torch.tensor(bench.THETA,dtype=torch.float64,device='cuda')
bufs={'loss': None}
# loss buffer with room for stamps: allocate bigger and slice
big = torch.zeros(B+64,dtype=torch.float64,device='cuda')
bufs={'loss': big[:B]}
for _ in range(3): o=mdl.oc_pdp_grad(u,th,dx,du,x0=x0,buffers=bufs)
torch.cuda.synchronize()
st = big[B:].view(torch.int64).cpu().numpy()
for blk in (0,1):
    s = st[blk*8:blk*8+5]; d = np.diff(s)
    print('block',[0,700][blk],'cycles: rollout+costate %d | terminal %d | backward %d | forward %d | total %d'%(d[0],d[1],d[2],d[3],s[4]-s[0]))

f = st[16:32]
print('backward step t=20: gather-issue %d | riccati_backward %d | gain stores %d | loop tail %d  (step total %d)' % (f[1]-f[0], f[2]-f[1], f[3]-f[2], f[4]-f[3], f[4]-f[0]))
print('forward  step t=20: loads+gathers %d | riccati_forward %d | acc/tail %d | (step total %d)' % (f[9]-f[8], f[10]-f[9], f[11]-f[10], f[12]-f[8]))
a = st[32:40]
print('phase totals over all chunks (cycles): eval_patha %d | costates(MFMA) %d | eval_pathb %d | riccati backward loop %d | eval_fwd %d | forward loop %d' % tuple(a[:6]))
print('rollout recursion alone (cycles): %d' % a[6])
