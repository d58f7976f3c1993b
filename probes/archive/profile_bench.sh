#!/bin/bash
# Profiles of the bench workload (run on the GPU box through gpurun; summaries land in gpurun_out/, copy the ones to keep into profiles/):
#   1. rocprofv3 --kernel-trace --stats            -> per-kernel durations (must agree with bench.py's roofline.kernel_ms)
#   2. rocprofv3 --kernel-trace --pmc FETCH_SIZE   -> HBM read traffic per dispatch   } separate passes, as the HBM section of
#   3. rocprofv3 --kernel-trace --pmc WRITE_SIZE   -> HBM write traffic per dispatch  } MI355X_MICROARCH.md prescribes
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o p -- $CMD > gpurun_out/prof_${TAG}.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_fetch -o p -- $CMD > gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_write -o p -- $CMD > gpurun_out/prof_${TAG}_write.log 2>&1
# 4. the same trace over the whole default bench (other_configs included): durations of every other kernel of the path
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_full -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${TAG}_full.log 2>&1
python - "$TAG" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {}
for name in ("fetch", "write"):
    vals = []
    for f in glob.glob("gpurun_out/prof_%s_%s/**/p_counter_collection.csv" % (tag, name), recursive=True):
        for r in csv.DictReader(open(f)):
            if "oc_pdp_fused" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    key = "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"
    out[key + "_KB_mean"] = sum(vals) / max(1, len(vals)); out[key + "_dispatches"] = len(vals)
# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of wide coalesced reads -> doubled; WRITE_SIZE as is
out["oc_pdp_fused_kernel_hbm_bytes_per_launch"] = 1024.0 * (2.0 * out["FETCH_SIZE_KB_mean"] + out["WRITE_SIZE_KB_mean"])
out["note"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs`, per dispatch of the fused OC kernel (oc_pdp_fused3_kernel by default, B=1024); FETCH_SIZE doubled (gfx950 correction of the guide), WRITE_SIZE uncorrected"
json.dump(out, open("gpurun_out/pmc_hbm_traffic_%s.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
import shutil
for f in glob.glob("gpurun_out/prof_%s/**/p_kernel_stats.csv" % tag, recursive=True):
    print(open(f).read())
    shutil.copy(f, "gpurun_out/%s_bench_kernel_stats.csv" % tag)
for f in glob.glob("gpurun_out/prof_%s_full/**/p_kernel_stats.csv" % tag, recursive=True):
    shutil.copy(f, "gpurun_out/%s_bench_full_kernel_stats.csv" % tag)
PY
