"""Where the runner / streamer lqrSolver kernel (lqr_solve_stream_kernel, -DPDP_LQS_TIMING build) spends its cycles: trajectory 0 at C3 sizes
(the workload of bench.py's C3_materialised_lqrSolver entry)."""
import sys, os, subprocess, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import codegen, runtime as rt, zoo
import bench
out = '/tmp/libpdp_hip_lqstiming.so'
EXTRA = [a for a in os.environ.get('PDP_EXTRA', '').split() if a]
subprocess.run([codegen.HIPCC] + codegen.CORE_FLAGS + EXTRA + ['-DPDP_LQS_TIMING', '-I', codegen.CSRC, os.path.join(codegen.CSRC, 'pdp_lqr.hip'), '-o', out], check=True)
rt.CORE_LIB = out
rt._core = None
lib = rt.load_core()
mdl = zoo.get('quadrotor', 'irl')
B, T = 1024, bench.HORIZON
x0, u, dx, du = bench.synth_inputs(B, 1000)
theta = torch.tensor(bench.THETA, dtype=torch.float64, device='cuda')
x, _ = mdl.oc_rollout(rt.dev(x0), rt.dev(u), theta)
lam = mdl.oc_costate(x, rt.dev(u), theta)
aux = mdl.oc_auxsys(x, rt.dev(u), lam, theta)
def run():
    return rt.lqr_solve(aux["dynF"], aux["dynG"], aux["Hxx"], aux["Huu"], aux["hxx"], aux["hxe"], E=aux["dynE"], Hxu=aux["Hxu"], Hxe=aux["Hxe"], Hue=aux["Hue"])
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [run() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
st = torch.zeros(8, dtype=torch.int64, device='cuda')
lib.pdp_lqs_read_stamps.restype = C.c_int
lib.pdp_lqs_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
lib.pdp_lqs_read_stamps(st.data_ptr(), rt.current_stream_ptr())
torch.cuda.synchronize()
s = st.cpu().numpy()
print('%.4f ms per call (incl. launch); trajectory 0, cycles: runner total %d (backward %d, forward %d) | runner waiting for the ring: backward %d, forward %d | streamer total %d, waiting for free slots %d'
      % (e0.elapsed_time(e1) / 10, s[0], s[1], s[0] - s[1], s[2], s[3] - s[2], s[4], s[5]))
