#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun; everything lands in gpurun_out/prof6, what is to be judged is copied into profiles/r06_*):
#   1. the default bench line, unprofiled, and ONE `rocprofv3 --kernel-trace --stats` run of it with bench.py's timing windows kept (PDP_BENCH_WINDOWS): the FULL per-kernel CSV
#      and probes/rocprof_match.py's table (every event-timed figure of the line against the dispatches of the same run);
#   2. the headline-only command under --kernel-trace --stats (the kernel whose average duration must agree with roofline.kernel_ms);
#   3. the PMC calibration (probes/pmc_calibrate.hip) and FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) over the bench command -> traffic.json,
#      which now records the digest of the kernel sources it was collected on (bench.py compares);
#   4. NEW: SQ counter passes of every latency-bound BASELINE kernel on its own (probes/floor_workloads.py) -> latency_floors.json: wave cycles, the part of them in which the
#      wave was issuing (SQ_ACTIVE_INST_ANY), parked on a wait (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY), instruction counts by kind - `floor_frac` of the bench entries.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof6
mkdir -p $O
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
if [ "$1" != "floors-only" ]; then
$BENCH > $O/plain.log 2>&1
grep '^{' $O/plain.log | tail -1 > $O/bench_line_unprofiled.json
PDP_BENCH_WINDOWS=$O/windows.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- $BENCH > $O/stats.log 2>&1
grep '^{' $O/stats.log | tail -1 > $O/bench_line_under_rocprof.json
python probes/rocprof_match.py $O/stats $O/windows.json $O/bench_line_unprofiled.json > $O/rocprof_match.txt 2>&1
echo "rocprof_match exit $?" >> $O/rocprof_match.txt
for f in $(find $O/stats -name 'p_kernel_stats.csv'); do cp $f $O/bench_full_kernel_stats.csv; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-scaling-configs > $O/stats_headline.log 2>&1
for f in $(find $O/stats_headline -name 'p_kernel_stats.csv'); do cp $f $O/bench_kernel_stats_two_streams.csv; done
grep '^{' $O/stats_headline.log | tail -1 > $O/bench_line_headline_under_rocprof.json
# the same command with every step on ONE stream: every dispatch of the headline kernel isolated, so that the AVERAGE of the stats summary is the duration of a launch
# (with the default two step streams a dispatch begins when it reaches the head of its queue, while the previous step still holds the CUs: begin to end it spans two steps)
PDP_BENCH_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_headline1 -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-scaling-configs > $O/stats_headline1.log 2>&1
for f in $(find $O/stats_headline1 -name 'p_kernel_stats.csv'); do cp $f $O/bench_kernel_stats.csv; done
grep '^{' $O/stats_headline1.log | tail -1 > $O/bench_line_headline_one_stream_under_rocprof.json
hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calibrate probes/pmc_calibrate.hip > $O/calib_build.log 2>&1
/tmp/pmc_calibrate 1024 4 > $O/calib_truth.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -o p -- /tmp/pmc_calibrate 1024 4 > $O/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib_write -o p -- /tmp/pmc_calibrate 1024 4 > $O/calib_write.log 2>&1
BENCH2="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-configs"
# (counter passes: bytes per launch do not depend on the clocks - without the 200 launches of load bench.py puts in front of every event-timed entry)
PDP_BENCH_NO_PREHEAT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $BENCH2 > $O/fetch.log 2>&1
PDP_BENCH_NO_PREHEAT=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $BENCH2 > $O/write.log 2>&1
python probes/pmc_traffic.py $O > $O/traffic_summary.txt 2>&1
tail -40 $O/rocprof_match.txt
tail -30 $O/traffic_summary.txt
fi
# ---- 4. SQ counters of the latency-bound kernels, each on its own
for w in sysid cp_poly cp_poly_c4 mlp oc_c4 headline solve solve_c2; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
             "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_F64"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/floor_${w}_$i -o p -- python probes/floor_workloads.py $w 6 > $O/floor_${w}_$i.log 2>&1
  done
done
python probes/latency_floors.py $O > $O/latency_floors.txt 2>&1
cat $O/latency_floors.txt | tail -60
