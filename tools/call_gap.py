#!/usr/bin/env python3
"""What lies BETWEEN two one-launch calls of the headline body (1 M-tet lattice, polar Jacobi FAST, reference threshold, 20 substeps per
call): frames issued back to back without synchronising, wall clock per frame against the call kernel's own duration, and -- from a
rocprofv3 trace of the same run -- the idle time between the end of one call kernel and the start of the next, with the copies in it.

    python tools/call_gap.py run [frames]                                   (prints wall clock per frame)
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/gap -- python tools/call_gap.py run
    python tools/call_gap.py parse gpurun_out/gap                           ->  profiles/r06_call_gap.txt
"""
import csv
import glob
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(frames):
    from tetsim_amd import SoftBodyHIP, make_lattice
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
    v, t = make_lattice(55)
    dt = (1.0 / 60.0) / 20
    for kw in (dict(ref_rotation_exit=True), dict()):
        b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", **kw)
        for _ in range(40):          # on the floor: every frame the same work
            b.simulateSubsteps(20, dt, pp)
        b.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(frames):
                b.simulateSubsteps(20, dt, pp)
            b.sync()
            best = min(best, (time.perf_counter() - t0) / frames)
        ev = sorted(b.timeSubsteps(20, dt, pp) for _ in range(9))[4]
        print("%-28s wall %.1f us per frame back to back = %.2f us per substep; one frame between two events %.1f us (path %d)" % (
            "reference threshold" if kw else "FAST exit", best * 1e6, best * 1e6 / 20, ev * 1e3, b.info.fused_particle_pass), flush=True)
        b.close()


def parse(d):
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for f in kt for r in csv.DictReader(open(f)))
    cs = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for f in mt for r in csv.DictReader(open(f)))
    calls = [k for k in ks if "pjb_call_kernel" in k[2]]
    gaps, inside = [], []
    for a, b2 in zip(calls, calls[1:]):
        g = (b2[0] - a[1]) / 1e3
        if g > 500.0:
            continue                 # (a synchronisation of the host lies in between)
        gaps.append(g)
        inside.append(sum(1 for c in cs if a[1] <= c[0] <= b2[0]) + sum(1 for k in ks if a[1] <= k[0] < b2[0] and k is not b2 and "pjb_call_kernel" not in k[2]))
    print("%d call kernels, %d back-to-back pairs: idle between the end of one and the start of the next  min %.1f  median %.1f  mean %.1f  p90 %.1f us; copies + other kernels in a gap: median %d" % (
        len(calls), len(gaps), min(gaps), statistics.median(gaps), statistics.mean(gaps), sorted(gaps)[int(0.9 * (len(gaps) - 1))], statistics.median(inside)))
    dur = [(k[1] - k[0]) / 1e3 for k in calls]
    print("call kernel duration median %.1f us; copies: %d, median duration %.2f us" % (statistics.median(dur), len(cs), statistics.median([(c[1] - c[0]) / 1e3 for c in cs]) if cs else 0.0))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    else:
        parse(sys.argv[2])
