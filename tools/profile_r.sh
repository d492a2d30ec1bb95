#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_r.sh <tag> [bench args]
# Produces gpurun_out/<tag>/{stats,pmc_fetch,pmc_write,pmc_sq}/... CSVs.
# Counter passes are separate from --kernel-trace --stats (gpurun rule) and
# from each other (TCC slot limits: FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=$1; shift
ARGS="$@"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 20 --warmup 2 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 20 --warmup 2 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -f csv -d $OUT/pmc_sq -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 20 --warmup 2 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_lds -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 20 --warmup 2 $ARGS > $OUT/pmc_lds.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line.json
# keep what tools/summarize_profile.py reads and stay under gpurun's 64 MiB merge limit: the per-kernel stats table and
# the balance kernel's rows of the counter tables (the traces also hold every torch kernel of the input generation)
find $OUT/stats -type f ! -name "run_kernel_stats.csv" -delete
for p in pmc_fetch pmc_write pmc_sq pmc_lds; do
  f=$OUT/$p/run_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep -E "balance_(pair_)?kernel" $f) > $f.tmp && mv $f.tmp $f; fi
  find $OUT/$p -type f ! -name "run_counter_collection.csv" -delete
done
du -sh $OUT
