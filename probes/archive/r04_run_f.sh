#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04f
for w in 8 12 16; do echo "== PDP_SYSID_GIVEN_WGS=$w (workgroups per CU the pool of the given-trajectory mode is sized for)"; PDP_SYSID_GIVEN_WGS=$w timeout 600 python probes/rollout_prepass.py 2>&1 | grep -v amdgpu.ids | grep "pre-pass 1\|default"; done > gpurun_out/r04f/prepass_wgs.txt
cat gpurun_out/r04f/prepass_wgs.txt
