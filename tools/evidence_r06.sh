#!/bin/bash
# Round 6: the hash-matched rocprofv3 passes bench.py reads roofline.traffic / roofline_valu from (tools/profile_r.sh: one
# --kernel-trace --stats pass, then separate --pmc passes; tools/profile_probe.sh for the batch-load probe), on the GPU box from the
# repo root.  Afterwards, here:
#   for t in gpurun_out/r06_*/; do python tools/summarize_profile.py $(basename $t); done; python tools/summarize_probe.py r06_batchload
# The tick profiles use the sweep entries' own protocol (50 steps after 20 warm-up launches + the device warm-up).
cd ${GRAFT_REPO_ROOT:-.}
for c in 2 3 4 5; do timeout 900 tools/profile_r.sh r06_cfg$c --config $c; done
timeout 900 tools/profile_r.sh r06_cfg5_n1 --config 5 --robots 2097152 --steps 20 --warmup 3
timeout 900 tools/profile_r.sh r06_tick_full65536 --config 3 --tick full --steps 50 --warmup 20
timeout 900 tools/profile_r.sh r06_tick_full262144 --config 3 --tick full --robots 262144 --steps 50 --warmup 20
timeout 900 tools/profile_r.sh r06_tick_fused4096 --config 2 --tick fused --steps 50 --warmup 20
timeout 900 tools/profile_r.sh r06_dense_cfg2 --config 2 --tune force_dense=1
timeout 900 tools/profile_r.sh r06_dense_cfg3 --config 3 --tune force_dense=1
timeout 900 tools/profile_r.sh r06_dense_tick_full65536 --config 3 --tick full --steps 50 --warmup 20 --tune force_dense=1
timeout 900 tools/profile_probe.sh r06_batchload
