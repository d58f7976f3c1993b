#!/bin/bash
# round-4 validation on the GPU box: full GPU suite, smoke, the default bench line, rocprofv3 kernel stats of the headline-only bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -o p -- python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/final/stats.log 2>&1
tail -3 gpurun_out/final/pytest_gpu.log; tail -2 gpurun_out/final/smoke.log; head -c 1500 gpurun_out/final/bench_line.json; echo; find gpurun_out/final/stats -name "*kernel_stats.csv" | head -1 | xargs head -5
