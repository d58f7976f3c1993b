#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun; everything lands in gpurun_out/prof5, what is to be judged is copied into profiles/r05_*):
#   1. ONE `rocprofv3 --kernel-trace --stats` run of the default bench line (headline + other_configs + scaling_configs at N = 1; the CPU baseline is skipped - it launches
#      nothing) with bench.py's timing windows kept (PDP_BENCH_WINDOWS): the FULL per-kernel CSV, and probes/rocprof_match.py's table that holds every event-timed
#      figure of that line against the dispatches rocprofv3 recorded in the same run.
#   2. the PMC calibration (probes/pmc_calibrate.hip): FETCH_SIZE / WRITE_SIZE against known byte counts in this repository's access shapes -> factors.
#   3. FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) over the same bench command -> HBM bytes per launch of the headline kernel, the
#      lqrSolver stream kernel, getAuxSys, the OC solver, scaled by the measured factors instead of the blanket x2 of rounds 1-4 -> traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof5
mkdir -p $O
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
$BENCH > $O/plain.log 2>&1
grep '^{' $O/plain.log | tail -1 > $O/bench_line_unprofiled.json
PDP_BENCH_WINDOWS=$O/windows.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $BENCH > $O/stats.log 2>&1
grep '^{' $O/stats.log | tail -1 > $O/bench_line_under_rocprof.json
python probes/rocprof_match.py $O/stats $O/windows.json $O/bench_line_unprofiled.json > $O/rocprof_match.txt 2>&1
echo "rocprof_match exit $?" >> $O/rocprof_match.txt
for f in $(find $O/stats -name 'p_kernel_stats.csv'); do cp $f $O/bench_full_kernel_stats.csv; done
# ---- the headline alone (the command whose one kernel `roofline.achieved` prices: its average duration in this summary must agree with roofline.kernel_ms)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-scaling-configs > $O/stats_headline.log 2>&1
for f in $(find $O/stats_headline -name 'p_kernel_stats.csv'); do cp $f $O/bench_kernel_stats.csv; done
grep '^{' $O/stats_headline.log | tail -1 > $O/bench_line_headline_under_rocprof.json
# ---- calibration
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate probes/pmc_calibrate.hip > $O/calib_build.log 2>&1
/tmp/pmc_calibrate 1024 4 > $O/calib_truth.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o p -- /tmp/pmc_calibrate 1024 4 > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o p -- /tmp/pmc_calibrate 1024 4 > $O/calib_write.log 2>&1
# ---- traffic of the bench kernels
BENCH2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-configs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $BENCH2 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $BENCH2 > $O/write.log 2>&1
python probes/pmc_traffic_r05.py $O > $O/traffic_summary.txt 2>&1
cat $O/rocprof_match.txt | tail -60
cat $O/traffic_summary.txt | tail -60
# ---- instruction counts of the tanh-MLP step at C5's total batch (8192 trajectories, [13, 13], T = 100): four trajectories per wavefront (default) against one (PDP_CP_MLP_VARIANT=3)
cat > /tmp/mlp_once.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from pdp_amd import runtime as rt, zoo
mdl = zoo.get("quadrotor", "oc")
rng = np.random.default_rng(0)
B, T, p = 8192, 100, 420
x0 = np.zeros((B, 13)); x0[:, :3] = rng.uniform(-2, 2, (B, 3)); x0[:, 6] = 1
x0d, thp, pol = rt.dev(x0), rt.dev(0.1 * rng.standard_normal(p)), rt.make_policy("mlp", layers=[13, 13, 4])
for _ in range(4):
    mdl.cp_step(pol, p, x0d, thp, T)
torch.cuda.synchronize()
PY
for v in 2 3; do
  PDP_CP_MLP_VARIANT=$v rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/mlp_pmc_$v -o p -- python /tmp/mlp_once.py > $O/mlp_pmc_$v.log 2>&1
done
python - <<'PY'
import csv, glob, json, collections
out = {}
for v, tag in ((2, "cp_step_mlp4t_kernel (four trajectories per wavefront, round 5 default)"), (3, "cp_step_mlp16_kernel (one trajectory per wavefront, PDP_CP_MLP_VARIANT=3)")):
    vals = collections.defaultdict(list)
    for f in glob.glob("gpurun_out/prof5/mlp_pmc_%d/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "cp_step_mlp" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[tag] = {k: sum(x[-3:]) / len(x[-3:]) for k, x in vals.items()}
out["note"] = "rocprofv3 --pmc, mean of the last three of four dispatches; C5 total batch: 8192 quadrotor trajectories, tanh MLP [13, 13] (p = 420), T = 100"
json.dump(out, open("gpurun_out/prof5/pmc_mlp_kernels.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
