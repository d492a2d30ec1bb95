"""Round 6 development tool: kernel time (HIP events, rotating cold-cache sets) of the complete tick and of the QP alone on the dense
12x12 form (force_dense = 1), 65 536 config-3 robots, for whatever library QC_LIB_PATH names.  usage: python tools/r06_dense_tick.py [race=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import json, subprocess
tune = [a for a in sys.argv[1:] if "=" in a]
def run(args):
    cmd = [sys.executable, "bench.py", "--config", "3", "--no-sweep", "--no-cpu-baseline", "--steps", "50", "--warmup", "20", "--tune", "force_dense=1"] + args
    for t in tune: cmd += ["--tune", t]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(os.path.dirname(__file__), ".."))
    d = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return d["roofline"]["avg_kernel_us"], d["solved_fraction"]
print("%-28s dense tick 65536: %.1f us (solved %.3f)   dense QP 65536: %.1f us" % ((os.environ.get("QC_LIB_PATH") or "in-tree") + " " + " ".join(tune), *run(["--tick", "full"]), run([])[0]), flush=True)
