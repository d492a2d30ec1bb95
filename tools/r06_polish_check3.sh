#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_polish; mkdir -p $O
timeout 900 python tools/r06_polish_stats.py 2>&1 | grep -v amdgpu.ids > $O/stats2.log
{
for i in 1 2 3; do
  for v in r05 p0 p1; do
    lib=""; tune=""
    [ $v = r05 ] && lib=tools/_build/libqc_r05.so
    [ $v = p0 ] && tune="--tune polish=0"
    QC_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 $tune 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg2', '$v', '%.4e QPs/s  %.3f us/step  kernel %.2f us' % (d['value'], d['ms_per_step'] * 1e3, d.get('roofline', {}).get('avg_kernel_us')))"
  done
done
for i in 1 2 3 4 5 6; do
  QC_LIB_PATH=tools/_build/libqc_r05.so timeout 600 python tools/cfg_times.py 1M 2M
  timeout 600 python tools/cfg_times.py 1M 2M
done
} 2>&1 | grep -v amdgpu.ids > $O/ab7.log
