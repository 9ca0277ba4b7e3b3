"""Development: the tiles' timelines against each other in the four-lane frame kernel (pj_quad.hip; ablation build, TETSIM_QUAD_POLL_DELAY=-1):
absolute s_memtime stamps of two consecutive substeps -- gather start, gather done, solve done, sum stored -- for every tile.  s_memtime has a different
offset per group of CUs (tiles fall into groups of ~4 that share one): compare tiles within a group only; the per-tile phase lengths
and periods are valid for all.     python tools/attic/frame_timeline.py [floor]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TETSIM_DEBUG_TRACE"] = "/tmp/frame_trace.bin"
os.environ["TETSIM_QUAD_POLL_DELAY"] = "-1"
os.environ.setdefault("TETSIM_HIP_LIB", os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_ablation.so"))
import numpy as np
from tetsim_amd import SoftBodyHIP
G = os.path.join(ROOT, "tests", "golden")
v = np.fromfile(os.path.join(G, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3); t = np.fromfile(os.path.join(G, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
if len(sys.argv) > 1 and sys.argv[1] == "floor":
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
dt = (1 / 60) / 20
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
for _ in range(6):
    b.simulateSubsteps(20, dt, pp)
b.sync(); b.close()
tr = np.fromfile("/tmp/frame_trace.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t0 = tr[:, 0].min()
r = tr - t0
print("tiles %d; cycles relative to the earliest gather start of substep n-3; columns: gather start | gather done | solve done | sum stored, twice" % len(tr))
order = np.argsort(r[:, 3])
for i in order:
    print("tile %2d  %6d %6d %6d %6d | %6d %6d %6d %6d   gather %5d solve %5d reduce %5d | period %5d" % ((i,) + tuple(r[i]) + (r[i, 1] - r[i, 0], r[i, 2] - r[i, 1], r[i, 3] - r[i, 2], r[i, 7] - r[i, 3])))
print("spread of 'sum stored' over the tiles: %d cycles (min %d, max %d); of 'gather done' of the NEXT substep: min %d max %d" % (r[:, 3].max() - r[:, 3].min(), r[:, 3].min(), r[:, 3].max(), r[:, 5].min(), r[:, 5].max()))
print("latest 'sum stored' -> earliest / median / latest 'gather done' of the next substep: %d / %d / %d cycles" % (r[:, 5].min() - r[:, 3].max(), np.median(r[:, 5]) - r[:, 3].max(), r[:, 5].max() - r[:, 3].max()))
