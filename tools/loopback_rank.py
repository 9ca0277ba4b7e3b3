"""What ONE interior rank of the slab decomposition costs per substep with the REAL RCCL kernels in the loop, on one GPU:
the middle slab of a 3-slab lattice (1 M tets + two ghost layers, two neighbours) whose halo partner is itself
(TETSIM_DEBUG_LOOPBACK_HALO=1: measurement only, the physics is meaningless).  Compared with the monolithic 1 M-tet body."""
import os, sys, time
os.environ["TETSIM_DEBUG_LOOPBACK_HALO"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("LOOPBACK_WITH_TORCH"):   # the bench's environment: torch's own streams and its RCCL communicator in the same process
    import torch, torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier()
from tetsim_amd import SoftBodyHIP, comm_info, comm_init, comm_unique_id, make_lattice, p2p_connect, p2p_export
PP = dict(gravity=0.0, friction=1000.0, density=1000.0, worldBounds=[-2.5, -1.0, -10.0, 2.5, 10.0, 10.0])   # g = 0: the self-halo keeps the lattice at rest
DT = (1 / 60) / 20
cells = 55
mv, mt = make_lattice(cells)
mono = SoftBodyHIP(mv, mt, None, dict(PP), solver="polar", precision="fast")
v, t = make_lattice(cells, nz=cells * 3)
plane = (cells + 1) ** 2
owner = np.minimum((np.arange(len(v)) // plane) // cells, 2).astype(np.int32)
owner[(np.arange(len(v)) // plane) >= 2 * cells] = 2
DEEP = bool(os.environ.get("LOOPBACK_DEEP"))   # a two-layer ghost region: ghosts cross every other substep (needs the peer-to-peer halo)
mid = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=3, part_index=1, vert_owner=owner, ref_fixed_bounds=False, deep_ghosts=DEEP)
comm_init(mid, comm_unique_id(), 0, 1)
if os.environ.get("LOOPBACK_P2P") or DEEP:   # the peer-to-peer halo instead of RCCL's grouped send/recv (the rank stores into its own ghost ranges)
    if not DEEP:
        mid.simulateSubsteps(2, DT, PP)
    p2p_connect(mid, [p2p_export(mid)])
TRANSPORT = ("peer-to-peer, two-layer ghosts" if DEEP else "peer-to-peer stores") if comm_info(mid)["p2p"] else "RCCL send/recv"
print("middle slab: %d owned particles, %d local tets (%d owned), %d neighbours" % (mid.info.owned_particles, mid.info.local_elems, mid.info.owned_elems, mid.info.num_neighbours))
pr = mid.profile(2, DT, PP)
print("interior tet kernel: %d of %d local tets (the rest, %.1f%%, are in halo-side tiles: they touch a ghost or a boundary particle)" % (
    pr["tets_per_tet_launch"], mid.info.local_elems, 100.0 * (1 - pr["tets_per_tet_launch"] / mid.info.local_elems)), flush=True)
CALLS, REPS = int(os.environ.get("LOOPBACK_CALLS", 50)), int(os.environ.get("LOOPBACK_REPS", 3))   # (small values: for a kernel trace)
for name, body in (("monolithic", mono), ("middle rank, loopback halo: " + TRANSPORT, mid)):
    if name == "monolithic" and os.environ.get("LOOPBACK_SKIP_MONO"): continue
    for _ in range(10): body.simulateSubsteps(20, DT, PP)
    body.sync()
    for rep in range(REPS):
        t0 = time.perf_counter()
        for _ in range(CALLS): body.simulateSubsteps(20, DT, PP)
        th = time.perf_counter() - t0
        body.sync(); tt = time.perf_counter() - t0
        print("%-48s host enqueue %.1f us, wall %.1f us per substep" % (name, th / (20 * CALLS) * 1e6, tt / (20 * CALLS) * 1e6), flush=True)
    assert np.isfinite(body.pos).all()
