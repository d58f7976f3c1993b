#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_examples.py -x -q -m gpu > gpurun_out/r04k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04k/pytest.log; tail -12 gpurun_out/r04k/pytest.log
timeout 300 python examples/sysid_pdp.py --system quadrotor --iters 300 --lr 1e-4 --graph 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline --no-scaling-configs > gpurun_out/r04k/bench.json 2> gpurun_out/r04k/bench.err; tail -2 gpurun_out/r04k/bench.err
