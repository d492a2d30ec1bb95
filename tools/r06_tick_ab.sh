#!/bin/bash
# Round 6: alternating A/B of the complete tick (bench.py --tick full, the sweep entries' protocol) between libraries.
# usage: tools/r06_tick_ab.sh <tag> <rounds> <lib ...>   ("" = in-tree)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
N=$2; shift; shift
LIBS=("$@")
{
for i in $(seq 1 $N); do
  for lib in "${LIBS[@]}"; do
    for nr in 65536 262144; do
      QC_LIB_PATH=$lib timeout 300 python bench.py --tick full --config 3 --robots $nr --no-sweep --no-cpu-baseline --steps 50 --warmup 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('tick %-7d %-28s kernel %.2f us  step %.2f us' % ($nr, '$lib' or 'in-tree', d['roofline']['avg_kernel_us'], d['ms_per_step'] * 1e3))"
    done
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/tick_ab.log
cat $O/tick_ab.log
