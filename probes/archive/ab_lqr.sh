for v in 2 1; do
  PDP_LQR_VARIANT=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_lqr_$v.json
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open("gpurun_out/ab_lqr_%s.json" % v)); c = d["other_configs"]["C3_materialised_lqrSolver_B1024"]
print("variant", v, c["kernel_ms"], c["traj_per_s"], c["achieved_gbps"], c["frac"])
PY
done
