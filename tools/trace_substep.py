"""Development: timeline of the one-launch substep (pjb_substep_kernel) -- the tiles' phases and the particle waves' behind them;
s_memtime stamps of the LAST launch of a short run, ablation build (python -m tetsim_amd.build --ablation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TETSIM_DEBUG_TRACE"] = "/tmp/substep_trace.bin"
os.environ.setdefault("TETSIM_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tetsim_amd", "libtetsim_hip_ablation.so"))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
v, t = make_lattice(55)
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
assert b.info.fused_particle_pass == 3
dt = (1 / 60) / 20
b.simulateSubsteps(20, dt, pp); b.sync()
if len(sys.argv) > 1 and sys.argv[1] == "single":
    for _ in range(3): b.simulate(dt, pp)          # the last launch is ONE substep
else:
    b.simulateSubsteps(20, dt, pp)                  # the last launch is 20 substeps: the rows hold the LAST substep's stamps
b.sync(); b.close()
tr = np.fromfile("/tmp/substep_trace.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
groups = (len(v) + 63) // 64
tiles, vg = tr[:-groups], tr[-groups:]
t0 = tiles[:, 0].min()   # (wall_clock64: one 100 MHz clock for the whole device)
print("tiles %d, particle groups %d; ticks are s_memtime units (100 MHz: 1 tick = 10 ns)" % (len(tiles), groups))
pc = lambda a: tuple(int(x) for x in np.percentile(a, [0, 10, 50, 90, 100]))
print("tile start            min %7d p10 %7d median %7d p90 %7d max %7d" % pc(tiles[:, 0] - t0))
print("tile end              min %7d p10 %7d median %7d p90 %7d max %7d" % pc(tiles[:, 6] - t0))
print("tile life             min %7d p10 %7d median %7d p90 %7d max %7d" % pc(tiles[:, 6] - tiles[:, 0]))
tn = ["start", "staged (wait + loads)", "after barrier", "solved", "stores issued", "after barrier 2", "end"]
for i in range(6):
    print("  tile %-32s median %7d p90 %7d max %7d" % (tn[i] + " -> " + tn[i + 1], np.median(tiles[:, i + 1] - tiles[:, i]), np.percentile(tiles[:, i + 1] - tiles[:, i], 90), (tiles[:, i + 1] - tiles[:, i]).max()))
names = ["wave start", "-", "all tiles' words seen", "particles finished"]
for i in (0, 2, 3):
    print("%-22s min %7d p10 %7d median %7d p90 %7d max %7d" % ((names[i],) + pc(vg[:, i] - t0)))
for i, j in ((0, 2), (2, 3)):
    print("  %-40s median %7d p90 %7d max %7d" % (names[i] + " -> " + names[j], np.median(vg[:, j] - vg[:, i]), np.percentile(vg[:, j] - vg[:, i], 90), (vg[:, j] - vg[:, i]).max()))
span = max(vg[:, 3].max(), tiles[:, 6].max()) - t0
for f in np.linspace(0, 1, 13):
    tt = t0 + f * span
    print("t=%7d  tiles started %5d finished %5d | particle waves started %5d waiting %5d finishing %5d done %5d" % (
        f * span, (tiles[:, 0] <= tt).sum(), (tiles[:, 6] <= tt).sum(), (vg[:, 0] <= tt).sum(),
        ((vg[:, 0] <= tt) & (vg[:, 2] > tt)).sum(), ((vg[:, 2] <= tt) & (vg[:, 3] > tt)).sum(), (vg[:, 3] <= tt).sum()))
