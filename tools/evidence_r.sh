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
timeout 300 python tools/race_scan.py 2>&1 | grep -v amdgpu.ids > $OUT/race_scan.log
{ timeout 600 python tools/clamp_scan_seeds.py 1 2 3 4 5 6 8 16; timeout 300 python tools/clamp_scan.py 1 2 4 5 6; } 2>&1 | grep -v amdgpu.ids > $OUT/clamp_scan.log
timeout 600 python tools/tail_race_scan.py 2>&1 | grep -v amdgpu.ids > $OUT/tail_race_scan.log
timeout 120 python tools/sort_stat.py 2>&1 | grep -v amdgpu.ids > $OUT/sort_stat.log
if [ -f tools/_build/libqc_balance_clk_all.so ]; then  # phase clocks averaged over every workgroup (tools/phase_clock.hip, -DQC_CLK_BLOCK=blockIdx.x)
  for a in "4096 2 4" "4096 2 4 0 race=0" "65536 3 1" "262144 3 1" "32768 3 2"; do
    echo "== tools/phase_clock.py $a"; QC_CLK_ALL=1 QC_CLK_LIB=tools/_build/libqc_balance_clk_all.so timeout 120 python tools/phase_clock.py $a 2>&1 | grep -v amdgpu.ids
  done > $OUT/phase_clock.log
fi
# the long parity / fuzz campaigns (tests/stress_*.py are scripts, not pytest cases)
{ timeout 900 python tests/stress_parity.py 1048576; timeout 300 python tests/stress_parity.py 40000; timeout 300 python tests/stress_fuzz.py;
  timeout 600 python tests/stress_fuzz.py 60 40000; timeout 300 python tests/stress_fuzz_states.py; timeout 300 python tests/stress_fuzz_tick.py;
  timeout 300 python tests/stress_fuzz_planner.py; } 2>&1 | grep -v amdgpu.ids > $OUT/stress.log
ls -la $OUT
