"""Round 6: the current-figures table at the top of DESIGN.md section 4, generated from profiles/r06_*.json (so that the document quotes
the profiles and not a transcription of them).  usage: python tools/design_table_r06.py  ->  markdown on stdout"""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROWS = [("cfg2", "config 2 (headline): 4 096 robots, cold"), ("cfg3", "config 3: 65 536, mixed contacts, cold"), ("cfg4", "config 4: 262 144, warm-started"),
        ("cfg5", "config 5 shard (the N = 8 kernel): 262 144, cold"), ("cfg5_n1", "config 5 on one GPU (N = 1 point): 2 097 152"),
        ("tick_full65536", "complete tick, clock running: 65 536"), ("tick_full262144", "complete tick: 262 144"), ("tick_fused4096", "fused tick: 4 096"),
        ("dense_cfg2", "dense W, config 2"), ("dense_cfg3", "dense W, config 3"), ("dense_tick_full65536", "dense W, complete tick: 65 536")]
print("| workload | kernel | rocprof avg µs | bench (HIP events) µs | bytes / robot | HBM frac (algorithmic) | counter ÷ algorithmic | FP64 issue | wave-instr / robot | waves / SIMD | LDS conflict |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for tag, what in ROWS:
    d = json.load(open(os.path.join(ROOT, "profiles", f"r06_{tag}.json")))
    k = max(d["kernels"], key=lambda r: float(r["TotalDurationNs"]))
    bl = d["bench_line"]; n = bl["config"]["robots_per_gpu"]
    avg = float(k["AverageNs"]); alg = d["algorithmic_bytes_per_launch"]; c = d["counters_per_launch_mean"]
    name = k["Name"].replace("void qc::", "").split("(")[0]
    print("| %s | `%s` | %.1f | %.1f | %.0f | %.4f | %.3f | %.2f | %.1f | %.2f | %.1f %% |" % (
        what, name, avg * 1e-3, bl["roofline"]["avg_kernel_us"], alg / n, alg / avg / 8000.0, d["traffic"]["ratio_to_algorithmic"], d["valu"]["issue_frac"],
        d["valu"]["insts_valu_per_robot"], d["valu"].get("resident_waves_per_simd", 0.0), 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1.0))))
p = json.load(open(os.path.join(ROOT, "profiles", "r06_batchload.json")))
print("| batch-load probe (no QP iterations): 2 097 152 | `%s` | %.1f | %.1f | 488 | %.4f | %.3f | - | %.1f | - | - |" % (
    p["kernel"]["name"].replace("void qc::", "").split("(")[0], p["kernel"]["avg_ns"] * 1e-3, p["roofline"]["bench_event_us"], p["roofline"]["algorithmic_frac"],
    p["traffic"]["ratio_to_algorithmic"], p["counters_per_launch_mean"].get("SQ_INSTS_VALU", 0) / p["bench_line"]["batch_load_probe"]["robots"]))
