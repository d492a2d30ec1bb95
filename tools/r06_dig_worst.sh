#!/bin/bash
# Round 6: the worst trials of a parameter campaign (tests/stress_fuzz.py), arbitrated by the LDP/KKT restatement (tools/fuzz_dig.py)
# usage: tools/r06_dig_worst.sh <seed> [trials=20000]
cd ${GRAFT_REPO_ROOT:-.}
S=$1; T=${2:-20000}
O=gpurun_out/r06_polish; mkdir -p $O
QC_FUZZ_SEED=$S timeout 900 python tests/stress_fuzz.py $T 2048 2>&1 | grep -v amdgpu.ids > $O/campaign_$S.log
tail -1 $O/campaign_$S.log
W=$(grep -E "^trial [0-9]+ (cold|warm)" $O/campaign_$S.log | awk '{for(i=1;i<=NF;i++) if($i=="err") print $(i+1), $2}' | sort -g -r | awk '!seen[$2]++' | head -4 | awk '{print $2}' | tr '\n' ' ')
echo "worst trials: $W"
QC_FUZZ_SEED=$S timeout 900 python tools/fuzz_dig.py $W 2>&1 | grep -v amdgpu.ids > $O/dig_$S.log
cat $O/dig_$S.log
