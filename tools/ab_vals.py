"""In-session sweep of an env knob over several values through bench.py:  python tools/ab_vals.py ENV v0 v1 v2 ..."""
import json, os, subprocess, sys
knob, vals = sys.argv[1], sys.argv[2:]
for rep in range(3):
    for val in vals:
        env = dict(os.environ); env[knob] = val
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("%s=%s value %.1f  ms/frame %.4f  tet %.2f us  vertex %.2f us  frac %.3f" % (knob, val, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["roofline"]["frac"]), flush=True)
