#!/bin/bash
# Round 6: the polish at acceptance - the small-w trial on every form, with and without it, and an alternating A/B of the
# throughput configs against round 5's library (tools/_build/libqc_r05.so, built from HEAD~ by hand).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_polish; mkdir -p $O
{
for t in "" force_dense=1 force_general=1 group=4 group=1 polish=0 polish=1 polish=3; do
  echo "== FUZZ_TUNE=$t"; QC_FUZZ_SEED=555 FUZZ_TUNE=$t timeout 600 python tools/fuzz_dig.py 138 2>&1 | grep -v "^   gpu   \|^   oracle \[" | head -8
done
} > $O/fuzz_dig_138.log 2>&1
{
for i in 1 2 3; do
  QC_LIB_PATH=tools/_build/libqc_r05.so timeout 600 python tools/cfg_times.py
  timeout 600 python tools/cfg_times.py
done
for i in 1 2 3; do
  for lib in tools/_build/libqc_r05.so ""; do
    QC_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg2', '$lib' or 'in-tree', d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_kernel_us'))"
  done
done
} > $O/ab.log 2>&1
