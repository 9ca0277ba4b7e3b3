"""Property tests (hypothesis) of the host-side schedules on arbitrary small meshes -- no GPU:
dependency levels, the clustered Gauss-Seidel schedule and the partition plans must be consistent for ANY connectivity,
including duplicated tets, isolated particles, one-tet meshes and wildly uneven ownership."""
import ctypes as C

import numpy as np
from hypothesis import given, settings, strategies as st

from tetsim_amd import _capi as capi
from tetsim_amd.partition import PartitionPlan


@st.composite
def meshes(draw):
    nv = draw(st.integers(4, 40))
    nt = draw(st.integers(1, 60))
    tets = []
    for _ in range(nt):
        tets.append(draw(st.lists(st.integers(0, nv - 1), min_size=4, max_size=4, unique=True)))
    return nv, np.array(tets, np.int32)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


@settings(max_examples=150, deadline=None)
@given(meshes())
def test_levels_keep_the_sequential_order(m):
    nv, t = m
    L = capi.lib()
    level = np.empty(len(t), np.int32)
    nl = C.c_uint32()
    assert L.tetsim_prep_levels(_ip(t.ravel()), len(t), nv, _ip(level), C.byref(nl)) == 0
    assert level.min() == 0 and level.max() == nl.value - 1
    last = {}
    for e in range(len(t)):
        need = max((last.get(int(v), -1) for v in t[e]), default=-1) + 1
        assert level[e] == need            # the earliest level after every earlier tet it shares a particle with
        for v in t[e]:
            last[int(v)] = level[e]


@settings(max_examples=150, deadline=None)
@given(meshes())
def test_cluster_schedule_is_a_valid_gauss_seidel_order(m):
    nv, t = m
    L = capi.lib()
    nt = len(t)
    order, launch, lane, step = (np.full(nt, -1, np.int32) for _ in range(4))
    nl, nc = C.c_uint32(), C.c_uint32()
    assert L.tetsim_prep_clusters(_ip(t.ravel()), nt, nv, _ip(order), _ip(launch), _ip(lane), _ip(step), C.byref(nl), C.byref(nc)) == 0
    assert sorted(order.tolist()) == list(range(nt))
    seq = t[order]
    last = {}
    for pos in range(nt):
        for v in seq[pos]:
            a = last.get(int(v))
            if a is not None:
                assert launch[a] < launch[pos] or (launch[a] == launch[pos] and lane[a] == lane[pos] and step[a] < step[pos])
            last[int(v)] = pos
    key = launch.astype(np.int64) * (1 << 32) + lane
    for k in np.unique(key):
        sel = key == k
        assert sel.sum() <= 8 and len(np.unique(seq[sel])) <= 8 and sorted(step[sel].tolist()) == list(range(sel.sum()))


@settings(max_examples=100, deadline=None)
@given(meshes(), st.integers(2, 5), st.randoms(use_true_random=False))
def test_partition_plans_fit_together(m, parts, rnd):
    nv, t = m
    owner = np.array([rnd.randrange(parts) for _ in range(nv)], np.int32)
    plans = [PartitionPlan(t, nv, parts, p, owner) for p in range(parts)]
    owned = np.concatenate([pl.local_to_global_vert[:pl.n_owned] for pl in plans])
    assert sorted(owned.tolist()) == list(range(nv))                       # every particle has exactly one owner
    assert sum(pl.n_owned_tets for pl in plans) == len(t)                 # every tet is "owned" once ...
    for p, pl in enumerate(plans):
        assert np.all(owner[pl.local_to_global_vert[:pl.n_owned]] == p)
        # ... and solved wherever it touches an owned particle
        touching = np.nonzero((owner[t] == p).any(axis=1))[0]
        assert pl.local_to_global_tet.tolist() == touching.tolist()
        assert np.array_equal(pl.local_to_global_vert[pl.local_tets], t[touching])
        ghosts = pl.local_to_global_vert[pl.n_owned:]
        assert np.all(owner[ghosts] != p)
        for nb in pl.neighbours:
            other = plans[nb.rank]
            back = [x for x in other.neighbours if x.rank == p]
            assert len(back) == 1
            # what we send is what the neighbour expects to receive, in the same order; and it is ours
            assert nb.send_global.tolist() == back[0].recv_global.tolist()
            assert np.all(owner[nb.send_global] == p)
            assert np.array_equal(pl.local_to_global_vert[nb.send_local], nb.send_global)
            assert np.array_equal(pl.local_to_global_vert[nb.recv_start:nb.recv_start + nb.recv_count], nb.recv_global)
        received = np.concatenate([nb.recv_global for nb in pl.neighbours]) if pl.neighbours else np.zeros(0, np.int32)
        assert sorted(received.tolist()) == sorted(ghosts.tolist())       # every ghost is refreshed by exactly one neighbour
