"""TETSIM_ORDER_CLUSTERED schedule (host_prep.cpp prep_clusters), checked on the CPU: the plan is a permutation, and any two
tets that share a vertex are solved in their sequential order -- by different launches, or by ONE lane in step order."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_mesh
from tetsim_amd import _capi as capi, make_lattice


def plan(t, nv):
    L = capi.lib()
    t = np.ascontiguousarray(t, np.int32)
    nt = len(t)
    out = [np.full(nt, -1, np.int32) for _ in range(4)]
    nl, nc = C.c_uint32(), C.c_uint32()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert L.tetsim_prep_clusters(ip(t.ravel()), nt, nv, *[ip(a) for a in out], C.byref(nl), C.byref(nc)) == 0
    return (*out, nl.value, nc.value)


def check(t, order, launch, lane, step):
    assert sorted(order.tolist()) == list(range(len(t)))
    seq = t[order]
    last = {}
    for pos in range(len(seq)):
        for v in seq[pos]:
            a = last.get(int(v))
            if a is not None:
                assert launch[a] < launch[pos] or (launch[a] == launch[pos] and lane[a] == lane[pos] and step[a] < step[pos]), (a, pos)
            last[int(v)] = pos
    # a lane's cluster: at most 8 tets over at most 8 vertices; steps are 0..n-1 without holes
    key = launch.astype(np.int64) * (1 << 32) + lane
    for k in np.unique(key):
        m = key == k
        assert m.sum() <= 8 and len(np.unique(seq[m])) <= 8
        assert sorted(step[m].tolist()) == list(range(m.sum()))


def test_lattice_cells_become_clusters():
    v, t = make_lattice(7)
    order, launch, lane, step, nl, nc = plan(t, len(v))
    check(t, order, launch, lane, step)
    assert nl == 8 and nc == 7 ** 3 and step.max() == 5   # one cluster per cell, 2x2x2 cell parities as colours


def test_dragon_and_random_meshes():
    v, t = load_mesh("dragon")
    order, launch, lane, step, nl, nc = plan(t, len(v))
    check(t, order, launch, lane, step)
    assert nl < 32 and nc < len(t) / 3
    rng = np.random.default_rng(7)
    for nv, nt in ((5, 1), (9, 40), (200, 900)):
        t = np.array([rng.choice(nv, 4, replace=False) for _ in range(nt)], np.int32)
        order, launch, lane, step, nl, nc = plan(t, nv)
        check(t, order, launch, lane, step)


def test_empty_mesh():
    order, launch, lane, step, nl, nc = plan(np.zeros((0, 4), np.int32), 0)
    assert nl == 0 and nc == 0
