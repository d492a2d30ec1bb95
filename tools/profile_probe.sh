#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_probe.sh <tag>      e.g. r06_batchload
# rocprofv3 passes of the batch-load probe (bench.py --probe-only: load -> assemble -> store of 2,097,152 robots, no QP iterations):
# one --kernel-trace --stats pass, then separate --pmc passes (gpurun rule; FETCH_SIZE and WRITE_SIZE do not fit one pass).
# Afterwards, here: python tools/summarize_probe.py <tag>  ->  profiles/<tag>.{md,json}, which bench.py's batch_load_probe reads.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --probe-only --steps 20"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o run -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o run -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o run -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY -f csv -d $OUT/pmc_sq -o run -- $CMD > $OUT/pmc_sq.log 2>&1
grep -h '"batch_load_probe"' $OUT/stats.log | tail -1 > $OUT/bench_line.json
find $OUT/stats -type f ! -name "run_kernel_stats.csv" -delete
for p in pmc_fetch pmc_write pmc_sq; do
  f=$OUT/$p/run_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep -E "balance_(pair_)?kernel" $f) > $f.tmp && mv $f.tmp $f; fi
  find $OUT/$p -type f ! -name "run_counter_collection.csv" -delete
done
du -sh $OUT
