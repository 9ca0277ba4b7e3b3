"""In-session A/B: reference formulation vs TETSIM_FLAG_CONSTANT_REST_SHAPE through bench.py (alternating runs)."""
import json, subprocess, sys
rows = []
for rep in range(3):
    for lean in (0, 1):
        cmd = [sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"] + (["--constant-rest-shape"] if lean else [])
        d = json.loads(subprocess.run(cmd, capture_output=True, text=True).stdout.strip().splitlines()[-1])
        rows.append((lean, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["roofline"]["frac"]))
        print("lean=%d value %.1f  ms/frame %.4f  tet %.2f us  vertex %.2f us  frac %.3f" % rows[-1], flush=True)
