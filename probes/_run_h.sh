export PDP_PROBE_CASES=1x20,4x20,1x20,4x20
for cfg in "4 0" "4 1" "4 2" "4 3" "8 3" "16 3"; do set -- $cfg
GPU_MAX_HW_QUEUES=$1 PDP_PROBE_DUMMY_STREAMS=$2 timeout 200 python probes/exchange_policies.py gpurun_out/r06_pol_q$1_d$2.json 2>&1 | grep -v amdgpu.ids | grep -v "stand-in\|headline kernel"
done
