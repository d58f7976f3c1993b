#!/bin/bash
# A/B of the poll interval of the runner / evaluator hand-over (PDP_MS2_SLEEP) in oc_solve_ms2_kernel: rebuilds the quadrotor model with each value (on the GPU box's
# copy of the tree only) and times the C3 solves of probes/predict_cost.py
cd $GRAFT_REPO_ROOT
for n in 1 4 8 2; do
  echo "== PDP_MS2_SLEEP=$n"
  PDP_HIP_EXTRA_FLAGS="-DPDP_MS2_SLEEP=$n" timeout 400 python probes/predict_cost.py 2>&1 | grep "^(1)\|^(2)\|(3'')"
done
