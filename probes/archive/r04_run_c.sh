#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04c
timeout 600 python probes/f3_variants_timing.py ahead2=probes/_build/libq_ahead2.so ahead3=probes/_build/libq_ahead3.so lanes=probes/_build/libq_lanes.so lanes+ahead3=probes/_build/libq_lanes_ahead3.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04c/variants.txt
for v in "-DPDP_F3_GAIN_AHEAD=3" "-DPDP_F3_ROLLOUT_LANES=1" "-DPDP_F3_ROLLOUT_LANES=1 -DPDP_F3_GAIN_AHEAD=3"; do
  echo "=== build flags: $v"
  PDP_EXTRA="$v" timeout 600 python probes/phase_timing3.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04c/stamps.txt 2>&1
cat gpurun_out/r04c/variants.txt gpurun_out/r04c/stamps.txt
