"""Throughput of the RCCL-mode halo path with N thread-ranks sharing ONE GPU (librccl replaced by tests/mock_rccl): every
rank has its own host thread, as in a real one-process-per-GPU run, so the host is not the bottleneck; the GPU executes the
work of all ranks.  Compare tets/s with the monolithic body: the difference is what the decomposition costs on the device
(ghost tets, split kernels, transfers).   TETSIM_RCCL_LIB=tests/mock_rccl/libmock_rccl.so python tools/attic/mock_ranks_throughput.py [N]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

os.environ["TETSIM_HALO_GRAPH"] = "0"   # the test double rendezvouses on the host: not capturable (the real RCCL is: tools/loopback_rank.py)
from tetsim_amd import SoftBodyHIP, comm_init, comm_unique_id, make_lattice
nranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cells = 55
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, worldBounds=[-2.5, -1.0, -10.0, 2.5, 10.0, 10.0])
DT = (1 / 60) / 20
v, t = make_lattice(cells, nz=cells * nranks)
plane = (cells + 1) ** 2
owner = np.minimum((np.arange(len(v)) // plane) // cells, nranks - 1).astype(np.int32)
mv, mt = make_lattice(cells)
mono = SoftBodyHIP(mv, mt, None, dict(PP), solver="polar", precision="fast")
for _ in range(5): mono.simulateSubsteps(20, DT, PP)
mono.sync(); t0 = time.perf_counter()
for _ in range(30): mono.simulateSubsteps(20, DT, PP)
mono.sync(); tm = (time.perf_counter() - t0) / 600
print("monolithic 1 M tets: %.1f us per substep -> %.1f G tet-solves/s" % (tm * 1e6, len(mt) / tm / 1e9), flush=True)
uid = comm_unique_id()
bar = threading.Barrier(nranks + 1)
def rank_main(r):
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=nranks, part_index=r, vert_owner=owner, ref_fixed_bounds=False)
    comm_init(body, uid, r, nranks)
    for _ in range(5): body.simulateSubsteps(20, DT, PP)
    body.sync(); bar.wait()
    for _ in range(30): body.simulateSubsteps(20, DT, PP)
    body.sync(); bar.wait()
    body.close()
ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
for th in ths: th.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); tt = (time.perf_counter() - t0) / 600
for th in ths: th.join()
print("%d thread-ranks x 1 M tets on one GPU: %.1f us per substep -> %.1f G tet-solves/s (%.0f%% of %d x monolithic)" %
      (nranks, tt * 1e6, len(t) / tt / 1e9, 100 * (tm * nranks) / tt, nranks), flush=True)
