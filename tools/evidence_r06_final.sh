#!/bin/bash
# Round 6, final tree: pytest -m gpu twice, the driver's bench command and the default one, the long fuzz campaigns from a fresh seed.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06_final; mkdir -p $O
for i in 1 2; do echo "== run $i: python -m pytest tests -m gpu -q"; python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -6; done > $O/pytest_gpu.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_flags.err | grep '^{' > $O/bench_driver_flags.json
python bench.py 2> $O/bench_default.err | grep '^{' > $O/bench_default.json
S=20260930
{
echo "== QC_FUZZ_SEED=$S"
echo "== stress_fuzz 20000 2048"; QC_FUZZ_SEED=$S timeout 900 python tests/stress_fuzz.py 20000 2048 2>&1 | tail -3
echo "== stress_fuzz_states 3000"; QC_FUZZ_SEED=$S timeout 600 python tests/stress_fuzz_states.py 3000 2>&1 | tail -2
echo "== stress_fuzz_tick 6000"; QC_FUZZ_SEED=$S timeout 900 python tests/stress_fuzz_tick.py 6000 2>&1 | tail -3
echo "== stress_fuzz_planner 400 2048 60"; QC_FUZZ_SEED=$S timeout 900 python tests/stress_fuzz_planner.py 400 2048 60 2>&1 | tail -3
echo "== stress_fuzz_gait 40 2048 600"; QC_FUZZ_SEED=$S timeout 900 python tests/stress_fuzz_gait.py 40 2048 600 2>&1 | tail -3
echo "== stress_parity 1048576"; timeout 900 python tests/stress_parity.py 1048576 2>&1 | tail -4
} 2>&1 | grep -v amdgpu.ids > $O/stress.log
