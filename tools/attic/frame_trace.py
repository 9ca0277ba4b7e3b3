"""Development: where a substep of the persistent frame kernel (pjb_frame_kernel) spends its cycles -- thread 0 of every tile adds up
s_memtime differences per phase over one call (ablation build; python -m tetsim_amd.build --ablation).
    python tools/attic/frame_trace.py [substeps] [floor]      floor: the Dragon starts 1 cm above the floor (contact: all nine rotation iterations)
TETSIM_QUAD=0: the one-lane-per-tet frame kernel on 256-tet tiles (pj_blocked.hip) instead of the four-lane one (pj_quad.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["TETSIM_DEBUG_TRACE"] = "/tmp/frame_trace.bin"
os.environ.setdefault("TETSIM_HIP_LIB", os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_ablation.so"))
import numpy as np
from tetsim_amd import SoftBodyHIP
G = os.path.join(ROOT, "tests", "golden")
v = np.fromfile(os.path.join(G, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3); t = np.fromfile(os.path.join(G, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2 and sys.argv[2] == "floor":
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
dt = (1 / 60) / 20
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
for _ in range(5):
    b.simulateSubsteps(n, dt, pp)
b.sync()
ms = min(b.timeSubsteps(n, dt, pp) for _ in range(10))
b.close()
tr = np.fromfile("/tmp/frame_trace.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
names = ["gather + particle update", "stage + barrier 1", "solve", "barrier 2", "reduce + store"]
print("frame kernel, Dragon, %d substeps per call: %.3f ms per call = %.2f us per substep (event timing of the whole call)" % (n, ms, ms / n * 1e3))
print("tiles %d; cycles per substep (s_memtime, thread 0 of each tile; sum over the call / n); polls = trips of the gather loop per substep" % len(tr))
per = tr[:, :5] / tr[:, 6:7]
for i, nm in enumerate(names):
    print("  %-26s median %7.0f  min %7.0f  max %7.0f" % (nm, np.median(per[:, i]), per[:, i].min(), per[:, i].max()))
print("  %-26s median %7.0f" % ("sum", np.median(per.sum(axis=1))))
print("  polls per substep: median %.2f max %.2f" % (np.median(tr[:, 5] / tr[:, 6]), (tr[:, 5] / tr[:, 6]).max()))
print("  XCC ids of the tiles' workgroups:", [int(x) for x in tr[:, 7]])
