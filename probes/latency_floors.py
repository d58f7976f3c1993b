#!/usr/bin/env python3
"""SQ counters of the latency-bound kernels (probes/profile_r06.sh, step 4) -> <dir>/latency_floors.json, copied to profiles/latency_floors.json, which bench.py reads for
`floor_frac`.  Per kernel (mean over the last four of six dispatches):
    wave_cycles            SQ_WAVE_CYCLES x 4 (the counter is in quad-cycles, MI355X_MICROARCH.md) summed over the launch's waves
    active / parked / stalled   SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (s_waitcnt, the LDS hand-over polls), SQ_WAIT_INST_ANY (issue stalls: a dependent MFMA / VALU result not
                           ready) - the guide: the three are disjoint and add up to the wave cycles
    floor_frac             active / wave cycles: the part of a wave's life in which it was ISSUING.  A kernel that runs one serial chain per wavefront with a SIMD (or half
                           of one) to itself cannot finish before its waves have issued their instructions: 1 / floor_frac is what removing every stall would buy.
    per wave:              instructions by kind, MFMA pipe busy cycles, the share of the launch's time the SIMDs were busy at all."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
KERNEL = {"sysid": "sysid_step_kernel", "cp_poly": "cp_step_poly2_kernel", "cp_poly_c4": "cp_step_poly2_kernel", "mlp": "cp_step_mlp4t_kernel", "oc_c4": "oc_pdp_fused3_kernel",
          "headline": "oc_pdp_fused3_kernel", "solve": "oc_solve_ms2_kernel", "solve_c2": "oc_solve_ms2_kernel"}
out = {}
for w, kern in KERNEL.items():
    vals = collections.defaultdict(list)
    names = collections.Counter()
    for f in sorted(glob.glob(os.path.join(O, "floor_%s_[0-9]" % w, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names[r["Kernel_Name"].split("(")[0].replace("void ", "")] += 1
    if not vals:
        out[w] = {"error": "no dispatch of %s recorded" % kern}
        continue
    c = {k: sum(v[-4:]) / len(v[-4:]) for k, v in vals.items()}
    waves = c.get("SQ_WAVES", 0) or 1
    wc = 4.0 * c.get("SQ_WAVE_CYCLES", 0)
    e = {"kernel": names.most_common(1)[0][0], "counters_mean_of_last_4_dispatches": c, "waves": waves,
         "wave_cycles_per_wave": wc / waves,
         "active_frac": 4.0 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None,
         "parked_on_waits_frac": 4.0 * c.get("SQ_WAIT_ANY", 0) / wc if wc else None,
         "issue_stall_frac": 4.0 * c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
         "instructions_per_wave": {k[len("SQ_INSTS_"):].lower() if k != "SQ_INSTS" else "all": c[k] / waves for k in sorted(c) if k.startswith("SQ_INSTS")},
         "mfma_pipe_busy_cycles_per_wave": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / waves,
         "active_by_unit_frac_of_wave_cycles": {k[len("SQ_ACTIVE_INST_"):].lower(): 4.0 * c[k] / wc for k in sorted(c) if k.startswith("SQ_ACTIVE_INST_") and wc}}
    e["floor_frac"] = e["active_frac"]
    out[w] = e
sys.path.insert(0, os.getcwd())
try:
    import time
    from pdp_amd import codegen
    out["collected"] = {"kernel_sources_sha1": codegen.kernel_sources_digest(), "date": time.strftime("%Y-%m-%d"), "by": "probes/profile_r06.sh"}
except Exception as ex:
    out["collected"] = {"error": repr(ex)}
json.dump(out, open(os.path.join(O, "latency_floors.json"), "w"), indent=1)
for w, e in out.items():
    if "floor_frac" in e:
        print("%-10s %-58s waves %5d  cycles/wave %9.0f  issuing %.3f  parked %.3f  issue-stalled %.3f  MFMA busy/wave %8.0f  instr/wave %s" %
              (w, e["kernel"][:58], e["waves"], e["wave_cycles_per_wave"], e["active_frac"], e["parked_on_waits_frac"], e["issue_stall_frac"], e["mfma_pipe_busy_cycles_per_wave"],
               {k: int(v) for k, v in e["instructions_per_wave"].items()}))
    else:
        print(w, e)
