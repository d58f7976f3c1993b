#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04g
for w in 8 12 16; do echo "== PDP_CP_GIVEN_WGS=$w"; PDP_CP_GIVEN_WGS=$w timeout 600 python probes/rollout_prepass_cp.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04g/prepass_cp.txt
cat gpurun_out/r04g/prepass_cp.txt
