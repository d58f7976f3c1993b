"""Headline kernel with P <- (P + P')/2 every k-th backward step instead of every step (-DPDP_F3_SYM_EVERY=k, k = 1, 2, 4): time at C3 / C4 shapes and the
deviation of loss / gradient from the k = 1 build and from the CPU oracle on a few samples.  Libraries: probes/_build/libf3_sym<k>_<system>.so, built by this
script when missing (hipcc) - build them beforehand in the container and they travel with the snapshot.
Since round 6 the switch is no longer in the product headers: apply probes/patches/retired_switches.patch to a scratch copy of the repository first."""
import sys, os, subprocess, numpy as np
sys.path.insert(0, os.getcwd())
from pdp_amd import codegen, zoo
os.makedirs('probes/_build', exist_ok=True)
libs = {}
for system in ("quadrotor", "rocket"):
    pb = zoo.make_problem(system, 'irl'); _, info = codegen.write_header(pb)
    for k in (1, 2, 4):
        out = 'probes/_build/libf3_sym%d_%s.so' % (k, system)
        if not os.path.exists(out):
            subprocess.run([codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + ['-DPDP_F3_SYM_EVERY=%d' % k, '-DPDP_MODEL_HEADER="generated/%s.h"' % info['name'],
                            '-I', codegen.CSRC, os.path.join(codegen.CSRC, 'pdp_model.hip'), '-o', out], check=True)
        libs[(system, k)] = out
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
import torch
from pdp_amd import runtime as rt, JinEnv
import bench
rng = np.random.default_rng(0)
cases = []
B = 1024
x0, u, dx, du = bench.synth_inputs(B, 1000)
cases.append(("quadrotor", 50, np.array(bench.THETA), x0, u, dx, du))
T = 100
x0 = np.zeros((B, 13)); x0[:, :3] = np.array([10, -8, 5.0]) + rng.standard_normal((B, 3)); x0[:, 3] = -0.1; x0[:, 6:10] = JinEnv.toQuaternion(1.5, [0, 0, 1])
u4 = np.tile(np.array([10.0, 0, 0]), (B, T, 1)) + 0.1 * rng.standard_normal((B, T, 3))
cases.append(("rocket", T, np.array([0.5, 1, 1, 1, 1, 1, 1, 50, 1, 1.0]), x0, u4, np.zeros((B, T + 1, 13)), np.zeros((B, T, 3))))
for system, T, th, x0, u, dx, du in cases:
    ref = None
    args = [rt.dev(a) for a in (u, th, dx, du)]
    x0d = rt.dev(x0)
    for k in (1, 2, 4):
        mdl = rt.ModelLib(libs[(system, k)])
        bufs = {}
        ms = bench._event_ms(torch, lambda: mdl.oc_pdp_grad(*args, x0=x0d, buffers=bufs), reps=30, warm=5)
        out = mdl.oc_pdp_grad(*args, x0=x0d)
        loss, grad = out["loss"].cpu().numpy(), out["grad"].cpu().numpy()
        if ref is None:
            ref = (loss, grad)
        dl = np.abs(loss - ref[0]).max() / np.abs(ref[0]).max()
        dg = (np.abs(grad - ref[1]).max(axis=1) / np.abs(ref[1]).max(axis=1)).max()
        print("%-9s T=%3d B=%d  symmetrise every %d step(s): %.4f ms   loss dev %.2e   gradient dev (per sample, relative to its largest entry, worst) %.2e" % (system, T, B, k, ms, dl, dg), flush=True)
