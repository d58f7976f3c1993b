"""Register / scratch use of the solver kernel instantiations of one zoo model built from the working tree (no GPU needed): python probes/ms_resources.py quadrotor [extra hipcc flags]"""
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
from pdp_amd import codegen, zoo
system = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
extra = sys.argv[2:]
pb = zoo.make_problem(system, 'irl'); _, info = codegen.write_header(pb)
os.makedirs('probes/_build', exist_ok=True)
out = 'probes/_build/libms_res_%s.so' % system
subprocess.run([codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + extra + ['-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'], '-I', codegen.CSRC,
                os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
for k, v in codegen.kernel_resources(out).items():
    if k.startswith("oc_solve_ms2"):
        print(system, k, {a: v[a] for a in ('vgpr', 'agpr', 'spill', 'scratch', 'sgpr')})
