#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04i
timeout 900 python -m pytest tests/test_gpu_examples.py -x -q -m gpu -k "graph or update" > gpurun_out/r04i/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04i/pytest.log; tail -15 gpurun_out/r04i/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-scaling-configs > gpurun_out/r04i/bench.json 2> gpurun_out/r04i/bench.err; tail -2 gpurun_out/r04i/bench.err
