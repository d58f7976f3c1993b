#!/bin/bash
# round 6: one GPU call that checks a solver build - parity tests of the solver, the phase stamps of the timing build, the C2 / C3 IRL-iteration entries of the bench line
# usage (on the GPU box, from the repository root): bash probes/r06_solver_check.sh <tag> [quick]
TAG=${1:-x}
mkdir -p gpurun_out
if [ "$2" != "quick" ]; then python -m pytest tests/test_gpu_ocsolver.py tests/test_gpu_predict.py -x -q 2>&1 | tail -3; fi
python probes/ms_phase_timing.py > gpurun_out/r06_ms2_phase_timing_$TAG.txt 2>&1
grep -v "^    it \|trajectory" gpurun_out/r06_ms2_phase_timing_$TAG.txt | grep -A3 "quadrotor warm B=1024\|cartpole warm"
python bench.py --steps 50 --no-scaling-configs --no-cpu-baseline > gpurun_out/r06_bench_$TAG.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r06_bench_$TAG.json"))
for k in ("C2_cartpole_irl_iteration_B256","C3_quadrotor_irl_iteration_B1024"):
    e=d["other_configs"][k]; print(k[:12], "iter %.4f solve %.4f unguarded %.4f grad %.4f cold %.3f loop %.4f plain-warm %.4f" % (e["kernel_ms"], e["oc_solve_ms"], e["oc_solve_ms_without_the_prediction_guard"], e["gradient_ms"], e["oc_solve_cold_ms"], e["irl_loop_wall_clock"]["ms_per_iteration_hipgraph_replay"], e["round3_pipeline_plain_warm_start"]["oc_solve_ms"]))
PY
