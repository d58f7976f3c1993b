#!/bin/bash
# SQ counter passes for the fused kernel (bench workload), 8 SQ counters per pass at most; summaries -> gpurun_out/pmc_sq_*.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-1024}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_sq_$i -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $B > gpurun_out/pmc_sq_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_sq_*/p_counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'oc_pdp_fused' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        agg[k] = sum(v) / len(v)
with open('gpurun_out/pmc_sq_summary.txt', 'w') as o:
    for k, v in agg.items():
        o.write('%-32s %16.1f\n' % (k, v)); print('%-32s %16.1f' % (k, v))
PY
