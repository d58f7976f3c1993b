#!/bin/bash
# round 4, GPU call A: primal prediction record (tests + cost probe at C3 and C2) and the cycle stamps of the headline kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_gpu_predict.py -x -q -m gpu > gpurun_out/r04a/pytest_predict.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04a/pytest_predict.log
timeout 600 python probes/predict_cost.py > gpurun_out/r04a/predict_cost_c3.txt 2>&1
timeout 600 python probes/predict_cost.py cartpole > gpurun_out/r04a/predict_cost_c2.txt 2>&1
timeout 600 python probes/phase_timing3.py > gpurun_out/r04a/phase_stamps.txt 2>&1
PDP_EXTRA="-DPDP_PHASE_TIMING_FINE" timeout 600 python probes/phase_timing3.py > gpurun_out/r04a/phase_stamps_fine.txt 2>&1
tail -3 gpurun_out/r04a/pytest_predict.log; cat gpurun_out/r04a/predict_cost_c3.txt gpurun_out/r04a/predict_cost_c2.txt gpurun_out/r04a/phase_stamps.txt gpurun_out/r04a/phase_stamps_fine.txt
