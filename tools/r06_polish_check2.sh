#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_polish; mkdir -p $O
timeout 900 python tools/r06_polish_stats.py > $O/stats.log 2>&1
{
for i in 1 2 3; do
  QC_LIB_PATH=tools/_build/libqc_r05.so timeout 600 python tools/cfg_times.py cfg3 cfg4 cfg5s 2M
  timeout 600 python tools/cfg_times.py cfg3 cfg4 cfg5s 2M polish=0
  timeout 600 python tools/cfg_times.py cfg3 cfg4 cfg5s 2M
done
for i in 1 2 3; do
  for v in r05 p0 p2; do
    lib=""; tune=""
    [ $v = r05 ] && lib=tools/_build/libqc_r05.so
    [ $v = p0 ] && tune="--tune polish=0"
    QC_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 $tune 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg2', '$v', d['value'], d['ms_per_step'], d.get('roofline', {}).get('avg_kernel_us'))"
  done
done
} > $O/ab2.log 2>&1
