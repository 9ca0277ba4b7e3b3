"""Development: frame times of the headline body frame by frame -- after creation, after 1 s of idling, and after 1 s of idling followed by
36 / 74 / 185 ms of device copies: what the driver's 5 warm-up + 20 timed frames sit on (profiles/archive/r03_warmup_transient.txt)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(55)
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast")
dt = (1 / 60) / 20
ts = []
for i in range(80):
    ts.append(b.timeSubsteps(20, dt, pp))
print("ms per frame, frames 0..79 (event-timed, one sync each):")
print(" ".join("%.3f" % x for x in ts))
time.sleep(1.0)
ts = [b.timeSubsteps(20, dt, pp) for _ in range(10)]
print("after 1 s idle:", " ".join("%.3f" % x for x in ts))
from tetsim_amd import measure_copy_bandwidth
for reps in (50, 150, 400):
    time.sleep(1.0)
    t0 = time.perf_counter(); bw = measure_copy_bandwidth(1 << 30, reps); el = time.perf_counter() - t0
    ts = [b.timeSubsteps(20, dt, pp) for _ in range(25)]
    print("after 1 s idle + %d x 1 GiB copy (%.0f ms, %.0f GB/s): first 5 warm-up frames %s | next 20: mean %.4f  [%s]" % (
        reps, el * 1e3, bw, " ".join("%.3f" % x for x in ts[:5]), sum(ts[5:]) / 20, " ".join("%.3f" % x for x in ts[5:])))
