"""Frame by frame: what a substep costs INSIDE the graphs (one graph replay of 20 substeps per frame, events around the replay, one
sync per frame) for the headline body with the FAST rotation exit and with the reference's threshold -- one body at a time (a second
body's traffic would push the first out of the Infinity Cache), `reps` bodies each, from rest.  The 1 M-tet lattice falls for ~19
frames, then lies on the floor.      python tools/attic/frame_series.py [frames=60] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(55)
dt = (1 / 60) / 20
out = {}
for rep in range(reps):
    for ref in (False, True):
        b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", ref_rotation_exit=ref)
        out.setdefault(ref, []).append([b.timeSubsteps(20, dt, pp) * 1e3 / 20 for _ in range(frames)])
        b.close()
fa, re = np.median(np.array(out[False]), axis=0), np.median(np.array(out[True]), axis=0)
print("us per substep inside the graph, frame by frame (median of %d bodies each, from rest)" % reps)
print("%-8s %10s %10s %8s" % ("frame", "FAST exit", "1e-9", "delta"))
for f in range(frames):
    print("%-8d %10.2f %10.2f %8.2f" % (f, fa[f], re[f], re[f] - fa[f]))
for a, b in ((5, 19), (25, frames)):
    print("frames %d..%d: FAST %.2f, 1e-9 %.2f" % (a, b, fa[a:b].mean(), re[a:b].mean()))
