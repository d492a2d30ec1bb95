#!/bin/bash
# Round 6: alternating A/B of the in-tree library against development builds under tools/_build/ (libqc_r05.so: round 5's HEAD
# built by hand) - throughput configs by HIP events (tools/cfg_times.py) and config 2 through bench.py.
# usage: tools/r06_ab.sh <tag> [rounds] [lib ...]      ("" = in-tree)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; mkdir -p $O
N=${2:-3}
shift; shift
LIBS=("$@")
[ ${#LIBS[@]} -eq 0 ] && LIBS=(tools/_build/libqc_r05.so "")
{
for i in $(seq 1 $N); do
  for lib in "${LIBS[@]}"; do QC_LIB_PATH=$lib timeout 600 python tools/cfg_times.py cfg3 cfg4 cfg5s 2M; done
done
for i in $(seq 1 $N); do
  for lib in "${LIBS[@]}"; do
    QC_LIB_PATH=$lib timeout 300 python bench.py --no-cpu-baseline --no-sweep --steps 200 --warmup 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('cfg2 %-28s %.4e QPs/s  %.3f us/step  kernel %.2f us' % ('$lib' or 'in-tree', d['value'], d['ms_per_step'] * 1e3, d.get('roofline', {}).get('avg_kernel_us')))"
  done
done
} 2>&1 | grep -v amdgpu.ids > $O/ab.log
