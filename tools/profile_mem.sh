#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/profile_mem.sh <tag> [bench args]
# Round 3: where a throughput launch waits - texture addresser / L1 (TCP) / L2 (TCC) / SQ wait counters of the
# balance kernel, one rocprofv3 --pmc pass per hardware block (counter slots per block are few), no tracing options.
set -u
TAG=$1; shift
ARGS="$@"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r CTRS; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS -f csv -d $OUT/p$i -o run -- python $ROOT/bench.py --no-cpu-baseline --no-sweep --steps 12 --warmup 2 $ARGS > $OUT/p$i.log 2>&1
  f=$OUT/p$i/run_counter_collection.csv
  if [ -f $f ]; then (head -1 $f; grep -E "balance_(pair_)?kernel" $f) > $f.tmp && mv $f.tmp $f; fi
  find $OUT/p$i -type f ! -name "run_counter_collection.csv" -delete 2>/dev/null
done <<LIST
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_WAIT_ANY
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_LEVEL_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU
LIST
python - <<PY
import csv, glob, collections, os
out = "$OUT"
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/run_counter_collection.csv")):
    per = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        per[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in per.items():
        v = v[2:] if len(v) > 4 else v  # drop the warm-up launches
        agg[k] = (sum(v) / len(v), len(v))
with open(out + "/summary.txt", "w") as fh:
    for k, (m, c) in agg.items():
        fh.write("%-40s %16.1f  (mean of %d launches)\n" % (k, m, c))
print(open(out + "/summary.txt").read())
PY
grep -h '"metric"' $OUT/p1.log | tail -1 > $OUT/bench_line.json
du -sh $OUT
