"""Condense a tools/profile_r.sh output directory (gpurun_out/<tag>) into
profiles/<tag>.md + profiles/<tag>.json (the files the judge reads).
Usage: python tools/summarize_profile.py <tag>"""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
K = "balance_kernel"
out = {"tag": tag}
stats = list(csv.DictReader(open(os.path.join(src, "stats", "run_kernel_stats.csv"))))
krows = [r for r in stats if K in r["Name"]]
out["kernels"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")} for r in krows]
counters = {}
res = {}
for p in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    f = os.path.join(src, p, "run_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            res = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
    for k, v in acc.items():
        counters[k] = sum(v) / len(v)
out["resources"] = res
out["counters_per_launch_mean"] = counters
# HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream.
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch_raw = counters["FETCH_SIZE"] * 1024.0
    out["traffic"] = {"fetch_bytes_raw": fetch_raw, "fetch_bytes_corrected_x2": 2 * fetch_raw,
                      "write_bytes": counters["WRITE_SIZE"] * 1024.0,
                      "hbm_bytes_per_launch": 2 * fetch_raw + counters["WRITE_SIZE"] * 1024.0}
bl = os.path.join(src, "bench_line.json")
if os.path.exists(bl) and os.path.getsize(bl):
    out["bench_line"] = json.loads(open(bl).read())
    rl = out["bench_line"]["roofline"]
    out["algorithmic_bytes_per_launch"] = rl["bytes_per_launch"]
    if "traffic" in out:
        out["traffic"]["ratio_to_algorithmic"] = out["traffic"]["hbm_bytes_per_launch"] / rl["bytes_per_launch"]
json.dump(out, open(os.path.join(dst, tag + ".json"), "w"), indent=1)
with open(os.path.join(dst, tag + ".md"), "w") as f:
    f.write(f"# rocprofv3 summary `{tag}`\n\n")
    f.write("Command: `tools/profile_r.sh %s ...` = `rocprofv3 --kernel-trace --stats -f csv` and then separate `--pmc` passes, each around `python bench.py --no-cpu-baseline --no-sweep ...`\n\n" % tag)
    f.write("## kernel stats (--kernel-trace --stats)\n\n| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
    for r in stats[:4]:
        f.write(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |\n")
    f.write("\n## resources (from the counter-collection CSV)\n\n" + json.dumps(res) + "\n\n## PMC counters (mean per launch of the balance kernel)\n\n| counter | value |\n|---|---|\n")
    for k, v in sorted(counters.items()):
        f.write(f"| {k} | {v:.1f} |\n")
    if "traffic" in out:
        t = out["traffic"]
        f.write("\n## HBM traffic per launch\n\nFETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a wide coalesced read).\n\n")
        f.write(f"* fetch (corrected): {t['fetch_bytes_corrected_x2']:.0f} B, write: {t['write_bytes']:.0f} B, total {t['hbm_bytes_per_launch']:.0f} B\n")
        if "ratio_to_algorithmic" in t:
            f.write(f"* algorithmic bytes per launch: {out['algorithmic_bytes_per_launch']} -> measured/algorithmic = {t['ratio_to_algorithmic']:.2f}\n")
    if "bench_line" in out:
        f.write("\n## bench line of the profiled run\n\n```\n" + json.dumps(out["bench_line"]) + "\n```\n")
print(open(os.path.join(dst, tag + ".md")).read())
