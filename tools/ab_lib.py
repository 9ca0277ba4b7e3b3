"""In-session A/B of two builds of libtetsim_hip.so through bench.py (alternating runs):  python tools/ab_lib.py libA.so libB.so"""
import json, os, subprocess, sys
libs = sys.argv[1:3]
for rep in range(3):
    for lib in libs:
        env = dict(os.environ); env["TETSIM_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "bench.py", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("%-28s value %.1f  ms/frame %.4f  tet %.2f us  vertex %.2f us  frac %.3f" % (os.path.basename(lib), d["value"], d["ms_per_step"], d["roofline"]["kernel_us"], d["roofline"]["vertex_kernel_us"], d["roofline"]["frac"]), flush=True)
