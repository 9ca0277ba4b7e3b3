#!/usr/bin/env python3
"""Would per-tet waits and early hand-overs INSIDE a cluster shorten the Gauss-Seidel chain?  A critical-path simulation on the product's own
schedule (tetsim_prep_clusters of a Kuhn lattice; CPU only, no resource limits): every tet solve takes tau, a particle handed from one
cluster to another is seen h later.  Model A = nh_call_kernel today (a cluster waits for its eight particles, solves, hands all on);
model B = a tet waits for its own four particles only and a particle leaves right after the last tet of the cluster that touches it.

    python tools/nh_dataflow_sim.py [cells]      (tau 0.45 us, h 1.0 / 1.8 / 2.6 us: profiles/r06_nh_link_trace.txt)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tetsim_amd import make_lattice
from tetsim_amd import _capi
L = _capi.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14
v, t = make_lattice(N)
t = np.ascontiguousarray(t, dtype=np.int32)
nt, nv = len(t), len(v)
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
order, launch, lane, step = [np.zeros(nt, np.int32) for _ in range(4)]
nl, nc = C.c_uint32(), C.c_uint32()
assert L.tetsim_prep_clusters(ip(t.ravel()), nt, nv, ip(order), ip(launch), ip(lane), ip(step), C.byref(nl), C.byref(nc)) == 0
print("lattice", N, "tets", nt, "colours", nl.value, "clusters", nc.value)
tets = t[order].tolist()
cid = (launch.astype(np.int64) * 10**7 + lane).tolist()
stp = step.tolist()
S = 16
def run(model, tau, h):
    avail = [0.0] * nv
    writer = [-1] * nv
    marks = []
    for s in range(S):
        i = 0
        while i < nt:
            j = i
            while j < nt and cid[j] == cid[i]: j += 1
            c = (s, cid[i])
            if model == "A":
                vs = set(x for k in range(i, j) for x in tets[k])
                t0 = max(avail[x] + (h if writer[x] != c else 0.0) for x in vs)
                t1 = t0 + tau * (j - i)
                for x in vs: avail[x] = t1; writer[x] = c
            else:
                te = 0.0
                for k in range(i, j):
                    ts = max(te, max(avail[x] + (h if writer[x] != c else 0.0) for x in tets[k]))
                    te = ts + tau
                    for x in tets[k]: avail[x] = te; writer[x] = c
            i = j
        marks.append(max(avail))
    return (marks[-1] - marks[S // 2 - 1]) / (S - S // 2)
for tau, h in ((0.45, 1.8), (0.45, 1.0), (0.45, 2.6)):
    print("tau %.2f h %.1f: model A (cluster waits for all, releases at end) %.1f us/substep; model B (per-tet waits, early release) %.1f us/substep" % (tau, h, run("A", tau, h), run("B", tau, h)))
