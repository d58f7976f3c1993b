"""Occupancy experiment on the headline kernel: build variants of the quadrotor OC library that fit TWO waves per
SIMD (LDS pool of PDP_FUSED_CHUNK = 14 steps: 20 KB per wave, amdgpu_waves_per_eu(2): 256 registers per wave) and time B = 1024 / 2048 /
4096 against the shipped one-wave-per-SIMD build.  `build` (run where hipcc is: here or on the GPU box) writes probes/variants/*.so,
`run` times them."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdp_amd import codegen, zoo
VARIANTS = {"base": [], "chunk14": ["-DPDP_FUSED_CHUNK=14"], "chunk14_w2": ["-DPDP_FUSED_CHUNK=14", "-DPDP_FUSED_WAVES=2"],
            "chunk14_w2_nopf": ["-DPDP_FUSED_CHUNK=14", "-DPDP_FUSED_WAVES=2", "-DPDP_FUSED_NO_PREFETCH"], "chunk25": ["-DPDP_FUSED_CHUNK=25"]}
OUT = os.path.join(ROOT, "probes", "variants")


def build():
    os.makedirs(OUT, exist_ok=True)
    pb = zoo.make_problem("quadrotor", "irl")
    _, info = codegen.write_header(pb)
    for name, extra in VARIANTS.items():
        out = os.path.join(OUT, "libq_%s.so" % name)
        cmd = [codegen.HIPCC] + codegen.HIP_FLAGS + codegen.OC_EXTRA_FLAGS + extra + ['-DPDP_MODEL_HEADER="generated/%s.h"' % info["name"], "-I", codegen.CSRC,
               os.path.join(codegen.CSRC, "pdp_model.hip"), "-o", out, "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        lines = r.stdout.splitlines()
        for i, l in enumerate(lines):
            if "Function Name" in l and "oc_pdp_fused" in l:
                keep = [x.split("remark:")[1].strip() for x in lines[i + 3:i + 12] if "remark:" in x and any(k in x for k in ("VGPRs:", "AGPRs", "Scratch", "Occupancy", "VGPRs Spill"))]
                print(name, "|", " ; ".join(keep))
        if r.returncode != 0:
            print(name, "FAILED", r.stdout[-2000:])


def run():
    import numpy as np, torch
    import bench
    from pdp_amd import runtime
    for name in VARIANTS:
        path = os.path.join(OUT, "libq_%s.so" % name)
        if not os.path.exists(path):
            continue
        mdl = runtime.ModelLib(path)
        th = torch.tensor(bench.THETA, dtype=torch.float64, device="cuda")
        res = []
        for B in (512, 1024, 2048, 4096):
            x0, u, dx, du = (torch.as_tensor(a, device="cuda") for a in bench.synth_inputs(B, 1000))
            bufs = {}
            for _ in range(3):
                o = mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in ev:
                a.record(); mdl.oc_pdp_grad(u, th, dx, du, x0=x0, buffers=bufs); b.record()
            torch.cuda.synchronize()
            ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
            res.append("B=%d %.3f ms (%.2f M/s)" % (B, ms, B / ms / 1e3))
            if B == 1024:
                g = o["grad"].cpu().numpy()
        print("%-16s %s  |grad| %.6e status %d" % (name, " | ".join(res), float(np.abs(g).sum()), int(o["status"].sum())))


if __name__ == "__main__":
    (build if sys.argv[1] == "build" else run)()
