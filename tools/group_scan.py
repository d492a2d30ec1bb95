"""Time the kernel for each lane-group width on configs 2/3/4 (dev tool)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
for cfg in (2, 3, 4):
    for g in (1, 2, 4):
        env = dict(os.environ)
        r = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-sweep", "--config", str(cfg), "--steps", "50", "--warmup", "5"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().split("\n")[-1])
            print(f"cfg{cfg} G={g}: {d['value']:.3e} QP/s  {d['ms_per_step']*1e3:.1f} us  solved {d['solved_fraction']}")
        except Exception as e:
            print("cfg", cfg, "G", g, "FAILED", r.stdout[-300:], r.stderr[-300:])
