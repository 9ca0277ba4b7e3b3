"""RCCL send/recv cost on this box (1 rank, self send/recv): eager vs captured in a HIP graph.  Run on a GPU host."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tetsim_amd import SoftBodyHIP, comm_init, comm_unique_id, make_lattice
from tetsim_amd import _capi as capi
v, t = make_lattice(4)
body = SoftBodyHIP(v, t, None, {}, solver="polar", precision="fast")
comm_init(body, comm_unique_id(), 0, 1)
L = capi.lib()
for nbytes in (4096, 197 * 1024, 1 << 20):
    for use_graph, per in ((0, 1), (1, 1), (1, 20)):
        h, tot = C.c_double(), C.c_double()
        rc = L.tetsim_comm_probe(body._h, nbytes, 400, use_graph, per, C.byref(h), C.byref(tot))
        msg = "" if rc == 0 else "  FAILED rc=%d: %s" % (rc, L.tetsim_last_error(body._h).decode())
        print("%8d B  %-14s host %6.1f us/group  total %6.1f us/group%s" % (nbytes, "eager" if not use_graph else "graph x%d" % per, h.value, tot.value, msg), flush=True)
