"""In-session A/B of a debug knob through bench.py (same box, alternating runs): python tools/ab_bench.py ENV_NAME"""
import json, os, subprocess, sys
knob = sys.argv[1]
rows = []
for rep in range(3):
    for val in ("0", "1"):
        env = dict(os.environ); env[knob] = val
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        rows.append((val, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"]))
        print(knob, "=", val, "value %.1f  ms/frame %.4f  tet %.2f us  vertex %.2f us" % rows[-1][1:], flush=True)
for val in ("0", "1"):
    r = [x for x in rows if x[0] == val]
    print(knob, "=", val, "median value %.1f, median tet kernel %.2f us" % (sorted(x[1] for x in r)[1], sorted(x[3] for x in r)[1]))
