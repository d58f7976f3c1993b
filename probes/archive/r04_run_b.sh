#!/bin/bash
# round 4, GPU call B: is the forward sweep of the headline kernel waiting for its gain loads?  stamps of three builds + bench timing of the 2-ahead build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04b
for v in "" "-DPDP_F3_GAIN_AHEAD=2" "-DPDP_F3_EXP_GAINS_STEP0"; do
  echo "=== build flags: ${v:-(shipped)}"
  PDP_EXTRA="$v" timeout 600 python probes/phase_timing3.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04b/gain_prefetch.txt 2>&1
cat gpurun_out/r04b/gain_prefetch.txt
