#!/bin/bash
# Round-3 profiles (run on the GPU box through gpurun; summaries land in gpurun_out/, the ones to keep are copied into profiles/):
#   1. rocprofv3 --kernel-trace --stats over the default bench (headline + other_configs + scaling_configs): per-kernel durations
#   2. HBM traffic of the headline kernel and of the OC solver kernel: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section)
#   3. SQ counters of the OC solver kernel (oc_solve_ms2_kernel), 8 per pass, on a script that only runs the C3 warm / cold solves
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof3
BENCH="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-scaling-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof3/stats -o p -- $BENCH > gpurun_out/prof3/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof3/fetch -o p -- $BENCH > gpurun_out/prof3/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof3/write -o p -- $BENCH > gpurun_out/prof3/write.log 2>&1
cat > /tmp/solve_only.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import zoo, runtime as rt
import bench
rng = np.random.default_rng(0)
B, T = 1024, 50
mdl = zoo.get("quadrotor", "irl")
th_star = np.array(bench.THETA)
x0 = rt.dev(bench.synth_inputs(B, 5)[0])
theta1 = rt.dev(th_star[None] * (1 + 0.02 * rng.uniform(-1, 1, (B, bench.N_PAR))))
demo = mdl.oc_solve_ms(x0, th_star, T)
warm = (demo["state"], demo["control"], demo["costate"])
for _ in range(4):
    mdl.oc_solve_ms(x0, theta1, T, warm=warm)
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/prof3/sq_$i -o p -- python /tmp/solve_only.py > gpurun_out/prof3/sq_$i.log 2>&1
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof3/sfetch -o p -- python /tmp/solve_only.py > gpurun_out/prof3/sfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof3/swrite -o p -- python /tmp/solve_only.py > gpurun_out/prof3/swrite.log 2>&1
python - <<'PY'
import csv, glob, json, collections, shutil
def mean_counter(pattern, kernel, last=None):
    vals = collections.defaultdict(list)
    for f in sorted(glob.glob(pattern, recursive=True)):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v[-last:]) / len(v[-last:]) if last else sum(v) / len(v), len(v)) for k, v in vals.items()}
out = {}
for tag, kern, pf, pw, last in (("oc_pdp_fused3_kernel", "oc_pdp_fused3", "gpurun_out/prof3/fetch/**/p_counter_collection.csv", "gpurun_out/prof3/write/**/p_counter_collection.csv", None),
                                ("oc_solve_ms2_kernel_warm", "oc_solve_ms2", "gpurun_out/prof3/sfetch/**/p_counter_collection.csv", "gpurun_out/prof3/swrite/**/p_counter_collection.csv", 4)):
    f = mean_counter(pf, kern, last).get("FETCH_SIZE", (0, 0)); w = mean_counter(pw, kern, last).get("WRITE_SIZE", (0, 0))
    # gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled; WRITE_SIZE as is; both in KB
    out[tag] = {"FETCH_SIZE_KB_mean": f[0], "WRITE_SIZE_KB_mean": w[0], "dispatches": f[1], "hbm_bytes_per_launch": 1024.0 * (2.0 * f[0] + w[0])}
out["oc_pdp_fused_kernel_hbm_bytes_per_launch"] = out["oc_pdp_fused3_kernel"]["hbm_bytes_per_launch"]
out["note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per dispatch; headline kernel on `python bench.py --steps 20 --warmup 3 "
               "--no-cpu-baseline --no-scaling-configs` (B = 1024), OC solver on four warm C3 solves (B = 1024, mean of those four dispatches); FETCH_SIZE doubled "
               "(gfx950 correction of the guide), WRITE_SIZE uncorrected")
json.dump(out, open("gpurun_out/prof3/pmc_hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
with open("gpurun_out/prof3/pmc_sq_solver.txt", "w") as o:
    o.write("SQ counters of oc_solve_ms2_kernel<quadrotor, TPW = 4>, mean over the last four dispatches (warm C3 solves, B = 1024, 3 iterations each), rocprofv3 --pmc, 8 counters per pass\n")
    for i in range(1, 5):
        for k, v in mean_counter("gpurun_out/prof3/sq_%d/**/p_counter_collection.csv" % i, "oc_solve_ms2", 4).items():
            o.write("%-32s %18.1f\n" % (k, v[0]))
print(open("gpurun_out/prof3/pmc_sq_solver.txt").read())
for f in glob.glob("gpurun_out/prof3/stats/**/p_kernel_stats.csv", recursive=True):
    shutil.copy(f, "gpurun_out/prof3/bench_kernel_stats.csv")
    print(open(f).read()[:6000])
PY
