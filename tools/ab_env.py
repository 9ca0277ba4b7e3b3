"""In-session A/B of an env knob through bench.py, alternating runs:  python tools/ab_env.py ENV_NAME [bench args...]"""
import json, os, subprocess, sys
knob, extra = sys.argv[1], sys.argv[2:]
for rep in range(3):
    for val in ("0", "1"):
        env = dict(os.environ); env[knob] = val
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("%s=%s value %.1f  ms/frame %.4f  tet %.2f us  vertex %.2f us  frac %.3f" % (knob, val, d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["roofline"]["frac"]), flush=True)
