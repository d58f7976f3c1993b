#!/bin/bash
# A/B the sched_group_barrier pipeline in the real bench build (no timing stamps): rebuild the quadrotor model with extra flags
for f in "" "-DPDP_SCHED_PIPELINE=4" "-DPDP_SCHED_PIPELINE=8" "-DPDP_SCHED_PIPELINE=12"; do
  rm -f pontryagin-differentiable-programming_amd/lib/libpdp_model_quadrotor_oc_*.so
  echo "== flags: $f"
  PDP_HIP_EXTRA_FLAGS="$f" python bench.py --no-cpu-baseline --steps 100 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3fM traj/s' % (r['ms_per_step'], r['value']/1e6))"
done
rm -f pontryagin-differentiable-programming_amd/lib/libpdp_model_quadrotor_oc_*.so
