#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e
timeout 900 python probes/rollout_prepass.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04e/prepass.txt; cat gpurun_out/r04e/prepass.txt
