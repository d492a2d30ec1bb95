"""-m gpu: bench.py keeps its contract (one JSON line on rank 0 with roofline + cpu_baseline at N = 1; the
torch.distributed.run launch with N > 1 ranks shards the batch and reduces the counters)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu(built):
    r = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-sweep"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 30 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["robots_per_gpu"] == 4096 and "workload" in d["config"] and d["solved_fraction"] == 1.0
    assert abs(d["value"] - 4096 * 30 / (d["ms_per_step"] * 30e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - 488 * 4096 / (rf["avg_kernel_us"] * 1e-6) / 1e9) < 1e-6 * rf["achieved"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "QPs/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 10 * cb["value"]


def test_bench_two_ranks_torchrun(built):
    """The N > 1 launch the driver uses, on this box's single GPU (QC_BENCH_ONE_DEVICE: both ranks on cuda:0, gloo
    for the barrier / counter reduction instead of RCCL)."""
    import socket

    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, QC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "20", "--warmup", "3", "--gather-results"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8192 and d["config"]["robots_per_gpu"] == 4096
    assert d["solved_fraction"] == 1.0 and d["value"] > 0
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    assert d["result_gather"]["bytes_per_rank"] == 4096 * 96 and d["result_gather"]["seconds"] > 0
