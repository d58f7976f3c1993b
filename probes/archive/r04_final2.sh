#!/bin/bash
# round-4 closing measurements: the default bench line and rocprofv3 kernel stats of the full bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final3
timeout 900 python bench.py > gpurun_out/final3/bench_line.json 2> gpurun_out/final3/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final3/stats -o p -- python bench.py --no-cpu-baseline > gpurun_out/final3/stats.log 2>&1
head -c 400 gpurun_out/final3/bench_line.json; echo; tail -3 gpurun_out/final3/bench.err
