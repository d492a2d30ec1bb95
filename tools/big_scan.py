"""Dev tool: refill threshold scan."""
import os, sys, subprocess, json
for cfg, n in ((4, 262144), (4, 1048576), (3, 1048576), (3, 262144)):
    for T in (2, 4, 8, 16, 32):
        env = dict(os.environ, QC_REFILL_T=str(T))
        cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--no-sweep", "--config", str(cfg), "--steps", "10", "--warmup", "2", "--robots", str(n)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            d = json.loads(r.stdout.strip().split("\n")[-1])
            print(f"cfg{cfg} n={n} T={T}: {d['value']:.3e} QP/s  {d['ms_per_step']*1e3:.1f} us")
        except Exception:
            print("FAILED", cfg, n, T, r.stderr[-300:])
