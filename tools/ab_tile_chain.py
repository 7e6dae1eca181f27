import json, subprocess, sys, os
for tc in ("1", "0"):
    env = dict(os.environ, SCVAE_TILE_CHAIN=tc)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "60", "--no-cpu-baseline", "--no-other-workloads"], env=env, capture_output=True, text=True)
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    print("tile chain", tc, r["ms_per_step"], r["step_ms_median"], r["value"])
