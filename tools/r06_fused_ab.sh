#!/bin/bash
# usage: tools/r06_fused_ab.sh <tag> <rounds> <lib ...>: the fused tick on config 2 (4096 robots: the chain-bound joint_q kernel)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
N=$2; shift; shift
LIBS=("$@")
{
for i in $(seq 1 $N); do
  for lib in "${LIBS[@]}"; do
      QC_LIB_PATH=$lib timeout 300 python bench.py --tick fused --config 2 --no-sweep --no-cpu-baseline --steps 200 --warmup 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('fused tick 4096 %-28s kernel %.2f us  step %.2f us' % ('$lib' or 'in-tree', d['roofline']['avg_kernel_us'], d['ms_per_step'] * 1e3))"
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/fused_ab.log
cat $O/fused_ab.log
