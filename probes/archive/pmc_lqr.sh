#!/bin/bash
# HBM traffic counters of the lqrSolver kernel at C3 sizes (bench.py's C3_materialised_lqrSolver workload): separate FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_lqr_$c -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_lqr_$c.log 2>&1
done
python - <<'PY'
import csv, glob, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("gpurun_out/pmc_lqr_%s/**/p_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "lqr_solve_stream_kernel" in r["Kernel_Name"] or "lqr_solve_kernel" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    out[c + "_KB_mean"] = sum(vals) / max(1, len(vals)); out[c + "_dispatches"] = len(vals)
out["hbm_bytes_per_launch"] = 1024.0 * (2.0 * out["FETCH_SIZE_KB_mean"] + out["WRITE_SIZE_KB_mean"])
out["note"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, dispatches of the lqrSolver kernel at C3 sizes (B=1024, T=50, with Lambda); FETCH_SIZE doubled (gfx950 correction of the guide)"
json.dump(out, open("gpurun_out/pmc_lqr_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
