"""Helper of test_gpu_polar.py: steps the MIDDLE slab of a 3-slab lattice whose halo partner is itself, through the REAL
RCCL send/recv kernels (TETSIM_DEBUG_LOOPBACK_HALO=1: liveness and determinism only, the physics is meaningless), and
prints a hash of the result.  Run once with TETSIM_HALO_GRAPH=0 (eager) and once with =1 (captured graph): same hash."""
import hashlib
import os
import sys

os.environ["TETSIM_DEBUG_LOOPBACK_HALO"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from tetsim_amd import SoftBodyHIP, comm_init, comm_unique_id, make_lattice  # noqa: E402

cells, precision = int(sys.argv[1]), sys.argv[2]
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, worldBounds=[-2.5, -1.0, -10.0, 2.5, 10.0, 10.0])
DT = (1 / 60) / 20
v, t = make_lattice(cells, nz=3 * cells, y0=0.02)
plane = (cells + 1) ** 2
owner = np.minimum((np.arange(len(v)) // plane) // cells, 2).astype(np.int32)
owner[(np.arange(len(v)) // plane) >= 2 * cells] = 2
body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, part_count=3, part_index=1, vert_owner=owner, ref_fixed_bounds=False)
comm_init(body, comm_unique_id(), 0, 1)
seq = (sys.argv[3] if len(sys.argv) > 3 else "7,7,20,-1,7,20,7x2,7x2,20").split(",")
for item in seq:   # the first call is always eager; repeated sizes replay cached graphs; -1 = one eager tetsim_step in between;
    n, _, f = item.partition("x")   # "7x2" = 7 substeps with dt * 2: the re-prediction hand-over in front of a replayed graph
    n, dt = int(n), DT * (float(f) if f else 1.0)
    if n < 0:
        body.simulate(dt, PP)
    else:
        body.simulateSubsteps(n, dt, PP)
    if os.environ.get("LOOPBACK_VERBOSE"):
        body.sync(); print("done", item, flush=True)
pos = body.pos
assert np.isfinite(pos).all()
print("HASH", hashlib.sha256(np.ascontiguousarray(pos, dtype="<f4").tobytes()).hexdigest()[:16], "ymin %.4f" % float(pos[:, 1].min()), flush=True)
body.close()
print("CLOSED", flush=True)
