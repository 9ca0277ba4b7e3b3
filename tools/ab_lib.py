"""In-session A/B of several builds of libtetsim_hip.so through bench.py (alternating runs, 3 rounds):
    python tools/ab_lib.py libA.so libB.so [libC.so ...] [-- extra bench args]
Per run: the value (and with the reference threshold), the tet kernel at nine iterations, with the FAST exit, on the floor (
nine rotation iterations everywhere), the particle kernel."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
libs = args[:args.index("--")] if "--" in args else args
for rep in range(3):
    for lib in libs:
        env = dict(os.environ, TETSIM_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs",
                              "--no-beyond-mall"] + extra, env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print("%-32s FAILED: %s" % (os.path.basename(lib), out.stderr[-300:]), flush=True)
            continue
        r = d["roofline"]
        fx, rt = r.get("fast_exit", r), r.get("timed_frames_reference_threshold", {})
        print("%-32s value %.1f (reference threshold %.1f)  ms/frame %.4f  tet: on the floor %.2f us (in graph ~%.2f), timed frames FAST exit %.2f us, with 1e-9 %.2f us  vertex %.2f us  frac %.3f" % (
            os.path.basename(lib), d["value"], d.get("value_reference_threshold", float("nan")), d["ms_per_step"], r["kernel_us"],
            r.get("on_floor", {}).get("in_graph", {}).get("kernel_us_implied", float("nan")), fx["kernel_us"], rt.get("kernel_us", float("nan")), r["vertex_kernel_us"], r["frac"]), flush=True)
