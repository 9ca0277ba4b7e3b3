"""Host-side view of the domain-decomposition plan (include/tetsim.h tetsim_plan_*); GPU-free.

The same C++ routine (csrc/host_prep.cpp build_partition) serves tetsim_create(part_count > 1), so a host that
runs its own transport -- or a CPU test -- sees exactly the numbering and halo lists the device path uses.
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def slab_owner(n_cells_xy, nz_cells, parts):
    """Vertex owner for an n x n x nz Kuhn lattice cut into `parts` z-slabs of whole vertex planes."""
    plane = (n_cells_xy + 1) * (n_cells_xy + 1)
    k = np.arange(plane * (nz_cells + 1)) // plane
    per = max(1, nz_cells // parts)
    return np.minimum(k // per, parts - 1).astype(np.int32)


def index_range_owner(num_particles, parts):
    """Equal contiguous index ranges: the cut a file's vertex order happens to give (the library's default until round 4; kept for tests
    that want ragged interfaces)."""
    return np.minimum(np.arange(num_particles, dtype=np.int64) * parts // max(num_particles, 1), parts - 1).astype(np.int32)


def partition(tetIds, num_particles, parts, vertices=None):
    """The library's built-in vertex partitioner (include/tetsim.h: tetsim_prep_partition): owner [num_particles] in [0, parts).
    vertices=None is what tetsim_create / tetsim_plan_create use when no owner is given (topology only); with coordinates the
    axis-aligned cuts are candidates too."""
    tets = np.ascontiguousarray(np.asarray(tetIds).reshape(-1), dtype=np.int32)
    v = None if vertices is None else np.ascontiguousarray(np.asarray(vertices).reshape(-1), dtype=np.float32)
    out = np.empty(int(num_particles), dtype=np.int32)
    capi.check(capi.lib().tetsim_prep_partition(v.ctypes.data_as(C.POINTER(C.c_float)) if v is not None else None, int(num_particles),
                                                tets.ctypes.data_as(C.POINTER(C.c_int32)), tets.size // 4, int(parts),
                                                out.ctypes.data_as(C.POINTER(C.c_int32))))
    return out


def partition_quality(tetIds, num_particles, parts, vert_owner=None):
    """Per part: owned / ghost / boundary particles, local / owned tets, neighbours (tetsim_prep_partition_quality), and the totals
    a decomposition is judged by.  vert_owner=None: the map the library picks itself."""
    tets = np.ascontiguousarray(np.asarray(tetIds).reshape(-1), dtype=np.int32)
    own = None if vert_owner is None else np.ascontiguousarray(vert_owner, dtype=np.int32)
    q = (capi.TetSimPartQuality * int(parts))()
    capi.check(capi.lib().tetsim_prep_partition_quality(tets.ctypes.data_as(C.POINTER(C.c_int32)), tets.size // 4, int(num_particles), int(parts),
                                                        own.ctypes.data_as(C.POINTER(C.c_int32)) if own is not None else None, q))
    per = [{k: int(getattr(q[r], k)) for k, _ in capi.TetSimPartQuality._fields_} for r in range(int(parts))]
    nt = tets.size // 4
    local = [p["local_elems"] for p in per]
    return {"parts": per,
            "ghost_particle_fraction": sum(p["ghost_particles"] for p in per) / max(1, sum(p["ghost_particles"] + p["owned_particles"] for p in per)),
            "ghost_tet_fraction": (sum(local) - nt) / max(1, sum(local)),
            "local_tet_imbalance": max(local) / (sum(local) / len(local)) - 1.0 if sum(local) else 0.0,
            "max_neighbours": max(p["num_neighbours"] for p in per)}


class Neighbour:
    __slots__ = ("rank", "send_local", "send_global", "recv_start", "recv_count", "recv_global", "contiguous",
                 "send2_local", "send2_global", "recv2_start", "recv2_count", "recv2_global")   # second ghost layer (depth 2)


class PartitionPlan:
    def __init__(self, tetIds, num_particles, part_count, part_index, vert_owner=None, depth=1):
        L = capi.lib()
        tets = np.ascontiguousarray(np.asarray(tetIds).reshape(-1), dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))  # noqa: E731
        owner = None if vert_owner is None else np.ascontiguousarray(vert_owner, dtype=np.int32)
        h = C.c_void_p()
        capi.check(L.tetsim_plan_create_deep(ip(tets), tets.size // 4, int(num_particles), part_count, part_index,
                                             ip(owner) if owner is not None else None, int(depth), C.byref(h)))
        try:
            sz = capi.TetSimPlanSizes()
            capi.check(L.tetsim_plan_sizes(h, C.byref(sz)))
            self.part_count, self.part_index = part_count, part_index
            self.n_owned, self.n_boundary = sz.owned_particles, sz.boundary_particles
            self.n_local, self.n_local_tets, self.n_owned_tets = sz.local_particles, sz.local_elems, sz.owned_elems
            self.local_to_global_vert = np.empty(self.n_local, dtype=np.int32)
            self.local_to_global_tet = np.empty(self.n_local_tets, dtype=np.int32)
            self.local_tets = np.empty(4 * self.n_local_tets, dtype=np.int32)
            capi.check(L.tetsim_plan_arrays(h, ip(self.local_to_global_vert), ip(self.local_to_global_tet), ip(self.local_tets)))
            self.local_tets = self.local_tets.reshape(-1, 4)
            # ghost layers (depth 2): ghosts [n_owned, n_owned + n_ghost1) share a tet with an owned particle, the rest with a
            # first-layer ghost; tet_layer 1 = a local tet that touches no owned particle
            self.depth = depth
            g1 = C.c_uint32()
            self.tet_layer = np.zeros(self.n_local_tets, dtype=np.uint8)
            capi.check(L.tetsim_plan_layers(h, C.byref(g1), self.tet_layer.ctypes.data_as(C.POINTER(C.c_uint8))))
            self.n_ghost1 = g1.value
            self.neighbours = []
            for i in range(sz.num_neighbours):
                r, c = C.c_int32(), C.c_int32()
                sc, rs, rc = C.c_uint32(), C.c_uint32(), C.c_uint32()
                capi.check(L.tetsim_plan_neighbour(h, i, C.byref(r), C.byref(sc), C.byref(rs), C.byref(rc), C.byref(c)))
                nb = Neighbour()
                nb.rank, nb.recv_start, nb.recv_count, nb.contiguous = r.value, rs.value, rc.value, bool(c.value)
                nb.send_local = np.empty(sc.value, dtype=np.int32)
                nb.send_global = np.empty(sc.value, dtype=np.int32)
                nb.recv_global = np.empty(rc.value, dtype=np.int32)
                capi.check(L.tetsim_plan_neighbour_ids(h, i, ip(nb.send_local), ip(nb.send_global), ip(nb.recv_global)))
                capi.check(L.tetsim_plan_neighbour_layer2(h, i, C.byref(sc), C.byref(rs), C.byref(rc)))
                nb.recv2_start, nb.recv2_count = rs.value, rc.value
                nb.send2_local = np.empty(sc.value, dtype=np.int32)
                nb.send2_global = np.empty(sc.value, dtype=np.int32)
                nb.recv2_global = np.empty(rc.value, dtype=np.int32)
                capi.check(L.tetsim_plan_neighbour_layer2_ids(h, i, ip(nb.send2_local), ip(nb.send2_global), ip(nb.recv2_global)))
                self.neighbours.append(nb)
        finally:
            L.tetsim_plan_destroy(h)
