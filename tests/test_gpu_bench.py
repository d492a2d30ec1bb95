"""-m gpu: bench.py keeps its contract (one JSON line on rank 0 with roofline + cpu_baseline at N = 1; `--gpus N`
starts N ranks by itself or checks the launcher's WORLD_SIZE; N > 1 shards config 5's fixed batch and reduces the
counters over the process group - RCCL when the box has the GPUs, gloo on one device otherwise)."""
import json
import os
import socket
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


def _free_port():
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


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
    # cold-cache protocol: rotating sets cover more than 512 MiB; the one-set replay is reported next to it
    assert "rotates through 269 distinct" in d["config"]["cache_protocol"]  # 512 MiB / (488 B x 4096) + 1
    assert d["warm_cache"]["value"] > 0 and d["warm_cache"]["avg_kernel_us"] > 0
    assert len(d["kernel_src_sha16"]) == 16


def test_bench_driver_line_with_the_whole_sweep(built):
    """The exact command the driver issues at round end (`bench.py --gpus 1 --steps 20 --warmup 5`, sweep included): one JSON
    line whose informational entries are all there - a failure anywhere in the sweep would cost the round its bench record."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    for k in CONTRACT + ("roofline", "cpu_baseline", "other_configs", "scaling_n1", "host_boundary", "batch_load_probe", "device_warmup", "warm_cache"):
        assert k in d, k
    oc = d["other_configs"]
    for k in ("config3", "config4", "config5_n1", "config5_shard8", "dense_config2", "dense_config3", "config2_fused_tick", "config3_full_tick", "full_tick_262144",
              "dense_full_tick_65536"):
        assert k in oc and oc[k]["solved_fraction"] == 1.0 and oc[k]["cold_cache"]["avg_kernel_us"] > 0, k
    # the tick with a general W has a number (the dense form's joint_q kernel, its clock running like the 6x6 tick's)
    dt = oc["dense_full_tick_65536"]
    assert dt["kernel"] == "dense-12x12" and dt["gait_clock"]["stance_to_swing_edges"] > 0 and dt["gait_clock"]["device_phase_vs_replayed_clock_max_abs"] == 0.0
    # the line ENDS with the compact summary (the driver stores the contract's keys and the last 2 000 characters)
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) < 1800
    sm = d["summary"]
    for k in ("cfg2", "cfg3", "cfg4_warm", "cfg5_shard8", "cfg5_n1", "dense2", "dense3", "fused_tick4096", "tick65536", "tick262144", "dense_tick65536", "batch_load"):
        assert sm[k]["us"] > 0 and 0 < sm[k]["hbm"] < 1, k
    assert sm["cfg5_n1"]["QPs"] == float("%.4g" % d["scaling_n1"]["value"]) and sm["sha"] == d["kernel_src_sha16"]
    # the N = 1 point of the scaling curve is named at the top level
    assert d["scaling_n1"]["robots"] == 2097152 and d["scaling_n1"]["value"] == oc["config5_n1"]["cold_cache"]["QPs_per_s"]
    # item 1: the complete tick's clock runs inside the timed region; the frozen-phase figure and a CPU baseline travel with it
    for k in ("config3_full_tick", "full_tick_262144"):
        e = oc[k]
        g = e["gait_clock"]
        assert g["stance_to_swing_edges"] > 0 and g["device_phase_vs_replayed_clock_max_abs"] == 0.0
        assert abs(e["bytes_per_robot"] - (1004 + 48 * g["edge_legs_per_robot_tick"])) < 1e-9
        assert e["frozen_phase"]["bytes_per_robot"] == 964 and e["frozen_phase"]["cold_cache"]["avg_kernel_us"] > 0
        assert e["cpu_baseline"]["kind"] == "port" and e["cpu_baseline"]["value"] > 0 and e["ticks_per_s"] > 20 * e["cpu_baseline"]["value"]
    # item 5: the dense form through the planner's choices - four racing lanes at config 2's size, one lane + LDS Hessian at config 3's
    assert oc["dense_config2"]["kernel"] == "dense-12x12" and oc["dense_config2"]["lanes_per_robot"] == 4
    assert oc["dense_config3"]["lanes_per_robot"] == 1 and oc["dense_config3"]["kernel_mode"] == 1 and oc["dense_config3"]["lds_bytes"] == 78 * 64 * 8
    assert oc["dense_config3"]["resident_workgroups"] >= 1024  # one workgroup per SIMD (two per CU before the LDS diet)
    assert d["batch_load_probe"]["frac"] > 0.4  # north_star: >= 40 % of HBM peak on the batch load
    bl = d["batch_load_probe"]  # ... with rocprofv3's kernel time and counters next to it whenever profiles/ holds passes of these kernel sources
    assert "traffic" in bl and (bl["traffic"] is None or (0.9 < bl["traffic_ratio_to_algorithmic"] < 1.3 and bl["traffic_source"].startswith("profiles/")))
    assert d["device_warmup"]["ms"] > 0 and d["device_warmup"]["warmup_launches_done"] >= 5
    # ... and the figure under rounds 1-4's protocol (W warm-up steps on a device that sat idle) travels next to `value`
    assert d["from_idle"]["idle_s"] == 2.0 and 0.4 * d["value"] < d["from_idle"]["value"] < 1.3 * d["value"]


def test_bench_complete_tick_advances_the_gait_clock(built):
    """`--tick full` times the COMPLETE tick - the on-device gait clock advances by 1/300 s per launch, so
    stance -> swing edges (foothold replanning + trajectory reset) happen inside the timed region at their natural rate; the line
    reports the realised rate, prices the roofline with it, and the phases the device left behind equal the replayed clock."""
    r = subprocess.run([sys.executable, "bench.py", "--tick", "full", "--config", "3", "--robots", "8192", "--steps", "24", "--warmup", "3",
                        "--no-sweep", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    g = d["gait_clock"]
    assert d["config"]["tick"] == "full" and abs(g["dt_s"] - 1 / 300) < 1e-15 and sum(g["timed_ticks_per_set"]) == 24
    assert g["stance_to_swing_edges"] > 0 and g["device_phase_vs_replayed_clock_max_abs"] == 0.0
    assert abs(g["edge_legs_per_robot_tick"] - g["expected_edge_legs_per_robot_tick"]) < 0.3 * g["expected_edge_legs_per_robot_tick"]
    per = 964 + 8 + 32 + 48 * g["edge_legs_per_robot_tick"]
    assert abs(d["roofline"]["bytes_per_launch"] - per * 8192) < 1e-6 * per * 8192
    assert d["solved_fraction"] == 1.0
    # the frozen-phase protocol of rounds 3-4 stays available
    r = subprocess.run([sys.executable, "bench.py", "--tick", "full-frozen", "--config", "3", "--robots", "8192", "--steps", "10", "--warmup", "3",
                        "--no-sweep", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert "gait_clock" not in d and d["roofline"]["bytes_per_launch"] == 964 * 8192


def _check_two_rank_line(d, total):
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == total
    assert d["config"]["robots_per_gpu"] == total // 2 and d["config"]["workload"].startswith("config5")
    assert d["solved_fraction"] == 1.0 and d["value"] > 0
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only
    ref = d["n1_reference"]  # the same batch on one GPU, measured in the same run (always, sweep or not)
    assert ref["robots"] == total and ref["solved_fraction"] == 1.0 and ref["value"] > 0
    assert abs(d["scaling_efficiency"] - d["value"] / (2 * ref["value"])) < 1e-12 and "configs[4]" in d["scaling_note"]
    rk = d["ranks"]
    assert 0 < rk["avg_kernel_us_min"] <= rk["avg_kernel_us_max"] and rk["allreduce_us"] > 0 and rk["backend"] in ("gloo", "nccl")
    _check_rccl_record(rk)
    assert d["from_idle"]["value"] > 0  # multi-rank lines carry the figure under rounds 1-4's protocol too


def _check_rccl_record(rk):
    """the N > 1 line says how the collective library moved its bytes."""
    rc = rk["rccl"]
    assert rc["backend"] == rk["backend"]
    if rc["backend"] == "nccl":
        for k in ("nranks", "channels", "via", "graph", "transport", "lines", "visible_devices", "peer_access"):
            assert k in rc, (k, rc)
        assert rc["visible_devices"] >= 1 and rc["log_bytes"] > 0, rc  # RCCL did write its INIT / GRAPH lines where we told it to


def test_bench_two_ranks_torchrun(built):
    """The N > 1 launch the driver uses (torch.distributed.run), on this box's single GPU (QC_BENCH_ONE_DEVICE: both
    ranks on cuda:0, gloo for the barrier / counter reduction instead of RCCL): config 5's fixed batch, sharded."""
    env = dict(os.environ, QC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2", "--gather-results",
           "--scaling", "strong", "--robots", "65536"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    _check_two_rank_line(d, 131072)
    assert d["result_gather"]["bytes_per_rank"] == 65536 * 96 and d["result_gather"]["seconds"] > 0


def test_bench_self_launch_two_ranks(built):
    """`python bench.py --gpus 2` without a launcher starts its own two ranks (one-device hook on a 1-GPU box)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--scaling", "strong", "--robots", "32768"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_two_rank_line(_last_json(r.stdout), 65536)


def test_bench_eight_ranks_before_the_eight_gpus(built):
    """The run the driver issues on an 8-GPU node - `python bench.py --gpus 8`, self-launched - executed with eight real
    ranks on this box's single GPU (QC_BENCH_ONE_DEVICE: all ranks on cuda:0, gloo for the barrier / counter reduction):
    the whole line must come out - global batch, eight contiguous equal shards, every robot solved, n1_reference measured
    by rank 0 while seven ranks wait under their own (long) limit, scaling_efficiency, per-rank kernel times, the gather."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", QC_BENCH_TIMEOUT_S="120")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "10", "--warmup", "2", "--scaling", "strong", "--robots", "16384",
                        "--gather-results"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    d = _last_json(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    total = 8 * 16384
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["global_batch"] == total and d["config"]["robots_per_gpu"] == 16384
    assert d["config"]["shards"] == [[k * 16384, (k + 1) * 16384] for k in range(8)]
    assert d["config"]["workload"].startswith("config5") and d["solved_fraction"] == 1.0
    assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    ref = d["n1_reference"]
    assert ref["robots"] == total and ref["solved_fraction"] == 1.0 and ref["value"] > 0
    assert abs(d["scaling_efficiency"] - d["value"] / (8 * ref["value"])) < 1e-12
    rk = d["ranks"]
    assert 0 < rk["avg_kernel_us_min"] <= rk["avg_kernel_us_max"] and rk["backend"] == "gloo" and rk["rccl"] == {"backend": "gloo"}
    assert d["result_gather"]["bytes_per_rank"] == 16384 * 96 and d["result_gather"]["seconds"] > 0
    assert "cpu_baseline" not in d and "roofline" in d


def test_bench_missing_rank_fails_fast(built):
    """A rank that never joins: the ranks that did start end with a one-line diagnostic and a non-zero exit code after
    QC_BENCH_TIMEOUT_S, instead of waiting out the collective library's own (ten-minute) timeout."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("QC_BENCH_FORCE_DIST",)}
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               QC_BENCH_ONE_DEVICE="1", QC_BENCH_TIMEOUT_S="10", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 60
    msg = [l for l in r.stderr.splitlines() if l.startswith("bench.py: rank 0/2")]
    assert len(msg) == 1 and "did not form within 10 s" in msg[0], r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # and no half-made result line


def test_bench_two_ranks_rccl_on_one_device(built):
    """N = 2 self-launch over RCCL itself with both ranks on cuda:0 (QC_BENCH_ONE_DEVICE_BACKEND=nccl) - what the 1-GPU
    box can show of the real N > 1 path.  RCCL may refuse two ranks of one communicator on one device ("Duplicate GPU
    detected"); then this is a skip and test_bench_two_ranks_rccl (two devices) stays the armed test."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(QC_BENCH_ONE_DEVICE="1", QC_BENCH_ONE_DEVICE_BACKEND="nccl", QC_BENCH_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1", "--scaling", "strong", "--robots", "32768"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        tail = (r.stderr + r.stdout)[-4000:]
        if "Duplicate GPU" in tail or "invalid usage" in tail.lower() or "did not form" in tail:
            pytest.skip("RCCL does not take two ranks on one device: " + tail.strip().splitlines()[-1][:200])
        assert False, tail
    d = _last_json(r.stdout)
    _check_two_rank_line(d, 65536)
    assert d["ranks"]["backend"] == "nccl"


def test_bench_one_rank_over_rccl(built):
    """RCCL itself on the 1-GPU box: a one-rank process group with backend "nccl" (QC_BENCH_FORCE_DIST), i.e. the
    init, the barriers around the timed region, the device-tensor all-reduces of the counters and the result gather
    all execute on the GPU - what N > 1 adds on top is only more ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("QC_BENCH_ONE_DEVICE",)}
    env.update(QC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--steps", "10", "--warmup", "2", "--config", "5",
           "--robots", "65536", "--no-sweep", "--no-cpu-baseline", "--gather-results"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 65536 and d["solved_fraction"] == 1.0
    assert d["result_gather"]["bytes_per_rank"] == 65536 * 96 and d["result_gather"]["seconds"] > 0
    _check_rccl_record(d["ranks"])
    assert d["ranks"]["rccl"]["transport"] == "none (one rank)" and d["ranks"]["rccl"]["peer_access"][0][0] == 1


def test_bench_two_ranks_rccl(built):
    """Two ranks on two GPUs over RCCL (backend "nccl"): config 5's full 2,097,152-robot batch, one shard per
    GPU.  Runs whenever the box has at least two devices; the driver's 1-GPU box skips it."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "QC_BENCH_ONE_DEVICE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "10", "--warmup", "2", "--gather-results"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    _check_two_rank_line(d, 2097152)
    assert d["result_gather"]["seconds"] > 0
    rc = d["ranks"]["rccl"]  # two devices: RCCL's own words about the transport
    assert rc["nranks"] == 2 and rc["channels"] and rc["transport"] != "unknown (no channel lines in the log)", rc
