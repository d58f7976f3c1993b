#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the bench kernels, scaled by factors MEASURED on known byte counts (probes/pmc_calibrate.hip), written to <dir>/traffic.json
(copied to profiles/traffic.json, which bench.py reads for roofline.traffic) and <dir>/pmc_calibration.json.  Run by probes/profile_r06.sh (round 5: profile_r05.sh) on the GPU box.
Round 6: the record says which kernel sources it was collected on (`collected.kernel_sources_sha1` = pdp_amd.codegen.kernel_sources_digest()); bench.py compares."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]


def counters(pattern, last=None):
    """{kernel name: [counter values per dispatch]} in KB (rocprofv3's unit for FETCH_SIZE / WRITE_SIZE)"""
    vals = collections.OrderedDict()
    for f in sorted(glob.glob(pattern, recursive=True)):
        for r in csv.DictReader(open(f)):
            vals.setdefault((r["Kernel_Name"], int(r["Grid_Size"])), []).append(float(r["Counter_Value"]))
    return vals


# ---- calibration: true bytes printed by the probe against the counters
truth = {}
for ln in open(os.path.join(O, "calib_truth.txt")):
    if ln.startswith("#") or not ln.strip():
        continue
    k, rd, wr = ln.split()
    truth[k] = (int(rd), int(wr))
cf = counters(os.path.join(O, "calib_fetch", "**", "*counter_collection.csv"))
cw = counters(os.path.join(O, "calib_write", "**", "*counter_collection.csv"))


def mean_of(c, kernel, pick=None):
    """mean counter value (bytes) over the dispatches of `kernel`; the first dispatch (cold caches) is dropped; pick = index among the grids seen (rows712 runs twice)"""
    keys = [k for k in c if k[0].split("(")[0].strip() == kernel]
    if not keys:
        return None
    key = keys[pick if pick is not None else 0]
    v = c[key][1:] if len(c[key]) > 1 else c[key]
    return 1024.0 * sum(v) / len(v)


cal = {}
for name in ("read16", "read8", "write8", "copy8_lds", "rows712", "rows712_big"):
    kern, pick = ("rows712", 0 if name == "rows712" else 1) if name.startswith("rows712") else (name, None)
    f, w_ = mean_of(cf, kern, pick), mean_of(cw, kern, pick)
    rd, wr = truth[name]
    cal[name] = {"true_read_bytes": rd, "true_write_bytes": wr, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w_,
                 "FETCH_SIZE_over_true_reads": (f / rd) if (f is not None and rd) else None, "WRITE_SIZE_over_true_writes": (w_ / wr) if (w_ is not None and wr) else None}
json.dump(cal, open(os.path.join(O, "pmc_calibration.json"), "w"), indent=1)
print(json.dumps(cal, indent=1))

# factors to MULTIPLY a counter by to get bytes, per access shape
fac = {}
if cal["read8"]["FETCH_SIZE_over_true_reads"]:
    fac["fetch_8B_per_lane_streaming"] = 1.0 / cal["read8"]["FETCH_SIZE_over_true_reads"]
if cal["read16"]["FETCH_SIZE_over_true_reads"]:
    fac["fetch_16B_per_lane_streaming (the guide's case)"] = 1.0 / cal["read16"]["FETCH_SIZE_over_true_reads"]
if cal["write8"]["WRITE_SIZE_over_true_writes"]:
    fac["write_8B_per_lane_buffer_store"] = 1.0 / cal["write8"]["WRITE_SIZE_over_true_writes"]
for k in ("rows712", "rows712_big"):
    if cal[k]["FETCH_SIZE_over_true_reads"] is not None:
        fac["fetch_gain_rows_written_then_reread (%s)" % k] = (1.0 / cal[k]["FETCH_SIZE_over_true_reads"]) if cal[k]["FETCH_SIZE_over_true_reads"] > 0 else None
    if cal[k]["WRITE_SIZE_over_true_writes"]:
        fac["write_gain_rows (%s)" % k] = 1.0 / cal[k]["WRITE_SIZE_over_true_writes"]
ff = fac.get("fetch_8B_per_lane_streaming", 2.0)
fw = fac.get("write_8B_per_lane_buffer_store", 1.0)

# ---- the bench kernels
bf = counters(os.path.join(O, "fetch", "**", "*counter_collection.csv"))
bw = counters(os.path.join(O, "write", "**", "*counter_collection.csv"))
out = {"calibration": {"factors_counter_to_bytes": fac, "applied": {"FETCH_SIZE": ff, "WRITE_SIZE": fw},
                       "how": "probes/pmc_calibrate.hip moves known byte counts (1 GiB buffers, four times the Infinity Cache) with this repository's access shapes - 8 B / lane "
                              "global loads, 8 B / lane range-checked buffer stores, an LDS-staged copy, and the gain-scratch pattern (712 B rows written, re-read by the same "
                              "wave); factor = true bytes / counter; rounds 1-4 applied a blanket x2 to FETCH_SIZE and x1 to WRITE_SIZE"},
       "kernels": {}}
for (name, grid), v in bf.items():
    if "pdp::" not in name and "pdp_" not in name:
        continue
    w_ = bw.get((name, grid))
    if not w_:
        continue
    f_mean, w_mean = 1024.0 * sum(v) / len(v), 1024.0 * sum(w_) / len(w_)
    key = "%s [grid %d]" % (name.split("(")[0].replace("void ", ""), grid)
    out["kernels"][key] = {"dispatches": len(v), "FETCH_SIZE_bytes_raw": f_mean, "WRITE_SIZE_bytes_raw": w_mean,
                           "hbm_bytes_per_launch_calibrated": ff * f_mean + fw * w_mean, "hbm_bytes_per_launch_round4_convention_x2_x1": 2.0 * f_mean + w_mean}
head = [k for k in out["kernels"] if "oc_pdp_fused3_kernel" in k and "grid 131072" in k and ("false" in k or ", 0>" in k or "0>" in k)]
if not head:
    head = [k for k in out["kernels"] if "oc_pdp_fused3_kernel" in k and "grid 131072" in k]
if head:
    best = max(head, key=lambda k: out["kernels"][k]["dispatches"])
    out["oc_pdp_fused_kernel_hbm_bytes_per_launch"] = out["kernels"][best]["hbm_bytes_per_launch_calibrated"]
    out["oc_pdp_fused_kernel_counted_as"] = best
sys.path.insert(0, os.getcwd())
try:
    import time
    from pdp_amd import codegen
    out["collected"] = {"kernel_sources_sha1": codegen.kernel_sources_digest(), "date": time.strftime("%Y-%m-%d"), "by": "probes/profile_r06.sh"}
except Exception as ex:
    out["collected"] = {"error": repr(ex)}
out["note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 20 --warmup 5 --no-cpu-baseline "
               "--no-scaling-configs` (probes/profile_r06.sh); counters in bytes (rocprofv3 reports KB), mean over all dispatches of a (kernel, grid) pair")
json.dump(out, open(os.path.join(O, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
