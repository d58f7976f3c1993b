"""Compile the quadrotor OC model with different hipcc flag sets (timing build) and report the fused kernel's cycle count per
trajectory for each (run on the GPU box: hipcc + GPU).  Usage: python probes/flag_sweep.py "flags A" "flags B" ..."""
import sys, os, subprocess, numpy as np, torch
sys.path.insert(0, os.getcwd())
from pdp_amd import codegen, zoo, runtime
import bench
pb = zoo.make_problem('quadrotor', 'irl'); _, info = codegen.write_header(pb)
B = 1024
x0, u, dx, du = (torch.as_tensor(a, device='cuda') for a in bench.synth_inputs(B, 1000))
th = torch.tensor(bench.THETA, dtype=torch.float64, device='cuda')
sets = sys.argv[1:] or [""]
for k, extra in enumerate(sets):
    out = '/tmp/libsweep%d.so' % k
    cmd = [codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + extra.split() + ['-DPDP_PHASE_TIMING', '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'],
                                                                                          '-I', codegen.CSRC, os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        print("%-70s COMPILE FAILED: %s" % (extra, r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "")); continue
    mdl = runtime.ModelLib(out)
    big = torch.zeros(B + 64, dtype=torch.float64, device='cuda')
    tot = []
    for _ in range(6):
        mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers={'loss': big[:B]}); torch.cuda.synchronize()
        st = big[B:].view(torch.int64).cpu().numpy(); tot.append(int(st[4] - st[0]))
    a = st[32:40]
    print("%-70s total %7d  (rollout %d costates %d riccati %d forward %d evals %d)" % (extra or "(baseline)", min(tot[2:]), a[6], a[1], a[3], a[5], a[0] + a[2] + a[4]))
