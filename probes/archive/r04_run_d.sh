#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_gpu_examples.py -x -q -m gpu -k "graph" > gpurun_out/r04d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04d/pytest.log
tail -25 gpurun_out/r04d/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-scaling-configs > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04d/bench.json"))
for k in ("C2_cartpole_irl_iteration_B256", "C3_quadrotor_irl_iteration_B1024"):
    e = d["other_configs"][k]
    print(k, e["ms"], e["prediction_record_kind"], json.dumps(e["irl_loop_wall_clock"]))
PY
tail -5 gpurun_out/r04d/bench.err
