#!/bin/bash
# A/B extra hipcc flags on the headline kernel (the model library is rebuilt on the box because the flag set is part of the build stamp)
for f in "" "-mllvm -amdgpu-mfma-vgpr-form"; do
  echo "== flags: $f"
  for i in 1 2; do PDP_HIP_EXTRA_FLAGS="$f" python bench.py --no-cpu-baseline --steps 100 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3fM traj/s' % (r['ms_per_step'], r['value']/1e6))"; done
done
