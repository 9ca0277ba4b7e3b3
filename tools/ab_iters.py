"""Iteration-count ablation of the blocked tet kernel through bench.py (development; wrong physics for iters != 9)."""
import json, os, subprocess, sys
for it in ("0", "3", "6", "9"):
    env = dict(os.environ); env["TETSIM_DEBUG_ITERS"] = it
    out = subprocess.run([sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print("iters=%s tet %.2f us  vertex %.2f us  ms/frame %.4f" % (it, d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["ms_per_step"]), flush=True)
