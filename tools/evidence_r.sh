#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/evidence_r.sh <tag>
# The development scans and microbenchmarks DESIGN.md quotes, into gpurun_out/<tag>/*.log (copy them to profiles/).
# Needs tools/_build/ubench_{mfma_reduce,valu,rsq} (hipcc --offload-arch=gfx950 -O3 tools/<name>.hip -o tools/_build/<name>).
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for u in ubench_mfma_reduce ubench_valu ubench_rsq; do
  [ -x tools/_build/$u ] && timeout 120 tools/_build/$u > $OUT/$u.log 2>&1
done
timeout 300 python tools/phase_split.py 2>&1 | grep -v amdgpu.ids > $OUT/phase_split.log
timeout 300 python tools/dense_scan.py 2>&1 | grep -v amdgpu.ids > $OUT/dense_scan.log
timeout 600 python tools/size_scan.py cold 2>&1 | grep -v amdgpu.ids > $OUT/size_scan_cold.log
timeout 600 python tools/size_scan.py warm 2>&1 | grep -v amdgpu.ids > $OUT/size_scan_warm.log
timeout 600 python tools/tick_scan.py 2>&1 | grep -v amdgpu.ids > $OUT/tick_scan.log
timeout 300 python tools/iter_scan.py 4096 2 2>&1 | grep -v amdgpu.ids > $OUT/iter_scan_cfg2.log
ls -la $OUT
