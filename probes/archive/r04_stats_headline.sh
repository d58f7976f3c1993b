#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline-only bench command (the kernel average that bench.py's HIP-event figure has to agree with)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/final2
CMD="python bench.py --no-other-configs --no-scaling-configs --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final2/stats -o p -- $CMD > gpurun_out/final2/stats.log 2>&1
tail -1 gpurun_out/final2/stats.log | head -c 600; echo
find gpurun_out/final2/stats -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-400
