"""Iteration-count ablation of the blocked tet kernel through bench.py.

Development only: runs the ABLATION build (python -m tetsim_amd.build --ablation -> libtetsim_hip_ablation.so), whose tet
kernel takes TETSIM_DEBUG_ITERS; the physics is wrong for iters != 9 and bench.py marks the line ("library.ablation": true).
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_ablation.so")
if not os.path.exists(LIB):
    raise SystemExit("build the ablation library first: python -m tetsim_amd.build --ablation")
for it in ("0", "3", "6", "9"):
    env = dict(os.environ, TETSIM_DEBUG_ITERS=it, TETSIM_HIP_LIB=LIB)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs", "--no-beyond-mall"],
                         env=env, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    assert d["library"]["ablation"] is True
    print("iters=%s tet %.2f us  (timed frames, FAST exit: %.2f us)  vertex %.2f us  ms/frame %.4f" % (it, d["roofline"]["kernel_us"], d["roofline"]["fast_exit"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["ms_per_step"]), flush=True)
