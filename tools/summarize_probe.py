"""Condense a tools/profile_probe.sh output directory (gpurun_out/<tag>) into profiles/<tag>.md + profiles/<tag>.json: the rocprofv3
evidence of north_star's "achieved HBM GB/s on the batch load" (bench.py's batch_load_probe reads the .json when its kernel-source
hash matches).  Usage: python tools/summarize_probe.py <tag>"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
line = json.loads(open(os.path.join(src, "bench_line.json")).read())
probe = line["batch_load_probe"]
stats = [r for r in csv.DictReader(open(os.path.join(src, "stats", "run_kernel_stats.csv"))) if "balance_" in r["Name"]]
# the probe's own launches: the process launches nothing else from this library
main = max(stats, key=lambda r: float(r["TotalDurationNs"]))
out = {"tag": tag, "kernel_src_sha16": line["kernel_src_sha16"], "bench_line": line,
       "kernel": {"name": main["Name"], "calls": int(main["Calls"]), "avg_ns": float(main["AverageNs"]), "min_ns": float(main["MinNs"]), "max_ns": float(main["MaxNs"])},
       "algorithmic_bytes_per_launch": probe["bytes"]}
counters = {}
for p in ("pmc_fetch", "pmc_write", "pmc_sq"):
    f = os.path.join(src, p, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    rows = [r for r in csv.DictReader(open(f)) if "balance_" in r["Kernel_Name"]]
    grids = collections.Counter(r["Grid_Size"] for r in rows)
    modal = grids.most_common(1)[0][0] if grids else None
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Grid_Size"] == modal:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            out["resources"] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
    for k, v in acc.items():
        counters[k] = sum(v) / len(v)
out["counters_per_launch_mean"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    # MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE count KiB; gfx950's FETCH_SIZE sees half of a wide coalesced read stream
    fetch_raw = counters["FETCH_SIZE"] * 1024.0
    hbm = 2 * fetch_raw + counters["WRITE_SIZE"] * 1024.0
    out["traffic"] = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2 * fetch_raw, "write_bytes": counters["WRITE_SIZE"] * 1024.0,
                      "hbm_bytes_per_launch": hbm, "ratio_to_algorithmic": hbm / probe["bytes"]}
avg_s = out["kernel"]["avg_ns"] * 1e-9
out["roofline"] = {"bound": "hbm", "peak_GBs": probe["peak"], "algorithmic_GBs": probe["bytes"] / avg_s / 1e9, "algorithmic_frac": probe["bytes"] / avg_s / 1e9 / probe["peak"],
                   "bench_event_us": probe["us"], "bench_event_frac": probe["frac"]}
if "traffic" in out:
    out["roofline"]["counter_GBs"] = out["traffic"]["hbm_bytes_per_launch"] / avg_s / 1e9
    out["roofline"]["counter_frac"] = out["roofline"]["counter_GBs"] / probe["peak"]
json.dump(out, open(os.path.join("profiles", tag + ".json"), "w"), indent=1)
with open(os.path.join("profiles", tag + ".md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`: the batch-load probe\n\n")
    f.write("Command: `tools/profile_probe.sh %s` = `rocprofv3 --kernel-trace --stats -f csv` and then separate `--pmc` passes, each around "
            "`python bench.py --probe-only --steps 20` (load -> assemble -> output transform -> store of %d robots, no QP iterations; kernel sources `%s`).\n\n"
            % (tag, probe["robots"], line["kernel_src_sha16"]))
    f.write("| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n| `%s` | %d | %.0f | %.0f | %.0f |\n\n"
            % (main["Name"][:110], out["kernel"]["calls"], out["kernel"]["avg_ns"], out["kernel"]["min_ns"], out["kernel"]["max_ns"]))
    r = out["roofline"]
    f.write("Algorithmic bytes per launch: %d (488 B x %d robots).  By rocprofv3's kernel average: **%.0f GB/s = %.1f %% of %.0f GB/s**; by HIP events inside "
            "bench.py (same run, under the profiler): %.1f us = %.1f %%.\n\n" % (probe["bytes"], probe["robots"], r["algorithmic_GBs"], 100 * r["algorithmic_frac"], r["peak_GBs"],
                                                                                 r["bench_event_us"], 100 * r["bench_event_frac"]))
    if "traffic" in out:
        t = out["traffic"]
        f.write("HBM traffic per launch (PMC, separate passes): FETCH_SIZE %.1f MB raw -> %.1f MB corrected x2 (gfx950 wide-read note, MI355X_MICROARCH.md), WRITE_SIZE %.1f MB: "
                "**%.1f MB = %.2fx the algorithmic bytes**, i.e. %.0f GB/s = %.1f %% of peak actually moved.\n\n"
                % (t["fetch_bytes_raw"] / 1e6, t["fetch_bytes_corrected_x2"] / 1e6, t["write_bytes"] / 1e6, t["hbm_bytes_per_launch"] / 1e6, t["ratio_to_algorithmic"],
                   r["counter_GBs"], 100 * r["counter_frac"]))
    if counters.get("SQ_INSTS_VALU"):
        f.write("SQ: %.3g VALU wave-instructions per launch (%.1f per robot), %.0f waves.\n" % (counters["SQ_INSTS_VALU"], counters["SQ_INSTS_VALU"] / probe["robots"], counters.get("SQ_WAVES", 0)))
print(open(os.path.join("profiles", tag + ".md")).read())
