"""Dev tool: group width / refill threshold at 1M robots."""
import os, sys, subprocess, json
for cfg in (3, 4):
    for g, T in ((1, 16), (1, 8), (1, 4), (2, 16), (2, 8), (2, 4), (2, 32)):
        env = dict(os.environ, QC_GROUP=str(g), QC_REFILL_T=str(T))
        cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--no-sweep", "--config", str(cfg), "--steps", "10", "--warmup", "2", "--n", "1048576"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().split("\n")[-1])
            print(f"cfg{cfg} n=1M G={g} T={T}: {d['value']:.3e} QP/s  {d['ms_per_step']*1e3:.1f} us")
        except Exception:
            print("FAILED", cfg, g, T, r.stderr[-300:])
