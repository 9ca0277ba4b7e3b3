"""Development: per-tile phase timeline of the blocked tet kernel (s_memtime stamps), last launch of a short run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["TETSIM_DEBUG_TRACE"] = "/tmp/tet_trace.bin"
# the stamps exist only in the ablation build (python -m tetsim_amd.build --ablation)
os.environ.setdefault("TETSIM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tetsim_amd", "libtetsim_hip_ablation.so"))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
v, t = make_lattice(55)
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
dt = (1 / 60) / 20
b.simulateSubsteps(20, dt, pp); b.sync()
for _ in range(3): b.simulate(dt, pp)
b.sync(); b.close()
tr = np.fromfile("/tmp/tet_trace.bin", dtype=np.uint64).reshape(-1, 8)
ts = tr[:, :7].astype(np.int64); hw = tr[:, 7]
t0 = ts[:, 0].min()
rel = (ts - t0)
print("tiles", len(tr), "kernel span (ticks)", rel[:, 6].max())
names = ["start", "loads landed(wave0)", "after barrier1", "solved", "stores issued", "after barrier2", "end"]
for i, n in enumerate(names):
    print("%-22s  min %8d  p10 %8d  median %8d  p90 %8d  max %8d" % ((n,) + tuple(int(x) for x in np.percentile(rel[:, i], [0, 10, 50, 90, 100]))))
d = np.diff(rel, axis=1)
for i in range(6):
    print("phase %-34s median %7d  p10 %7d  p90 %7d" % (names[i] + " -> " + names[i + 1], np.median(d[:, i]), np.percentile(d[:, i], 10), np.percentile(d[:, i], 90)))
# start-time histogram: how many workgroups are resident over time
starts, ends = np.sort(rel[:, 0]), np.sort(rel[:, 6])
span = rel[:, 6].max()
for f in np.linspace(0, 1, 11):
    tt = f * span
    print("t=%7d  started %5d  finished %5d  resident %5d" % (tt, (starts <= tt).sum(), (ends <= tt).sum(), (starts <= tt).sum() - (ends <= tt).sum()))
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; xcc = 0
print("distinct hw (cu,sh,se) ids:", len(np.unique(hw & 0xfff00)))
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "tet_trace.npy"), tr) if os.path.isdir("gpurun_out") else None
