#!/usr/bin/env python3
"""How many of the polar tet kernel's nine rotation iterations still move anything (SoftbodyGPU.js:122-139).

Development, ablation build only (python -m tetsim_amd.build --ablation): the kernel logs every tet's |omega|^2 per iteration
(pj_blocked.hip: pjb_log_iterations) and the library appends one line of counters per body to $TETSIM_DEBUG_ITER_HIST when
the body is destroyed.  Reported, for exit thresholds 1e-9 (the reference's), 1e-7, 3e-7 and 1e-6 on the CORRECTION iterations
2..9 (iteration 1 keeps 1e-9): the iterations a tet, and its wave (the loop is wave-uniform), would execute; and the
distribution of |omega| per iteration.

    python tools/attic/rotation_iterations.py > profiles/r04_rotation_iterations.txt      (on the GPU box)
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_ablation.so")
if not os.path.exists(LIB):
    raise SystemExit("build the ablation library first: python -m tetsim_amd.build --ablation")
os.environ["TETSIM_HIP_LIB"] = LIB
os.environ["TETSIM_FRAME_KERNEL"] = "0"          # small bodies: the per-substep kernels carry the log
THR = ["1e-9", "1e-7", "3e-7", "1e-6"]

import numpy as np  # noqa: E402


def parse(path):
    rows = []
    with open(path) as f:
        for line in f:
            a = line.split()
            rows.append((int(a[0]), int(a[1]), np.array(a[2:], dtype=np.float64)))
    return rows


def table(h, title):
    print(title)
    cnt = h[:80].reshape(2, 4, 10)
    for lvl, name in ((0, "tets "), (1, "waves")):
        tot = cnt[lvl, 0].sum()
        if tot == 0:
            continue
        print("  %s executing j iterations (%% of %d)      j=1     2     3     4     5     6     7     8     9   mean" % (name, tot))
        for k in range(4):
            row = cnt[lvl, k, 1:] / tot * 100.0
            mean = (cnt[lvl, k, 1:] * np.arange(1, 10)).sum() / tot
            print("    exit at |omega| < %-5s                       " % THR[k] + " ".join("%5.1f" % x for x in row) + "  %5.2f" % mean)
    om = h[80:].reshape(9, 22)
    print("  |omega| per iteration, %% of tets per half decade (column = upper edge)")
    edges = ["<1e-10"] + ["%.0e" % 10 ** ((b - 20) / 2.0) for b in range(1, 22)]
    keep = [b for b in range(22) if om[:, b].sum() > 0]
    print("    iter " + " ".join("%7s" % edges[b] for b in keep))
    for j in range(9):
        tot = om[j].sum()
        if tot:
            print("    %4d " % (j + 1) + " ".join("%7.2f" % (om[j, b] / tot * 100.0) for b in keep))
    print()


def bench_workload():
    from tetsim_amd import SoftBodyHIP, make_lattice
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
    v, t = make_lattice(55)
    dt = (1 / 60) / 20
    path = tempfile.mktemp(prefix="iter_hist_")
    os.environ["TETSIM_DEBUG_ITER_HIST"] = path
    marks = [1, 2, 3, 5, 10, 25, 50]
    for frames in marks:
        b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
        for _ in range(frames):
            b.simulateSubsteps(20, dt, pp)
        b.sync()
        b.close()
    rows = parse(path)
    os.unlink(path)
    prev, prev_f = np.zeros(278), 0
    print("== bench workload: 55^3-cell lattice (998,250 tets) falling from rest, 20 substeps per frame; cumulative counters of bodies run for", marks, "frames, differenced ==\n")
    for frames, (_, _, h) in zip(marks, rows):
        table(h - prev, "frames %d-%d" % (prev_f + 1, frames))
        prev, prev_f = h, frames
    table(rows[marks.index(25)][2], "frames 1-25 (the driver's 5 warm-up + 20 timed frames)")


def golden_cases():
    print("== GLSL-golden cases (tests/test_gpu_polar_reference.py, FAST blocked, whole recorded run each) ==\n")
    for name in ["lat4", "dragon", "dragon_grab", "lat4_drag", "hub", "lat12"]:
        path = tempfile.mktemp(prefix="iter_hist_")
        env = dict(os.environ, TETSIM_DEBUG_ITER_HIST=path, TETSIM_RECORD_ERRORS=os.devnull)
        subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_polar_reference.py"), "-m", "gpu", "-q", "-p", "no:cacheprovider",
                        "-k", "test_device_tracks_the_reference_glsl and %s-fast-False" % name], env=env, capture_output=True, text=True, cwd=ROOT)
        if not os.path.exists(path):
            print(name, ": no counters (test did not run?)\n")
            continue
        rows = parse(path)
        os.unlink(path)
        h = sum(r[2] for r in rows)
        table(h, "%s (%d tets)" % (name, rows[0][0]))


if __name__ == "__main__":
    bench_workload()
    golden_cases()
