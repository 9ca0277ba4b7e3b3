"""The .tetsim mesh container from Python: thin ctypes wrappers over tetsim_mesh_* in
include/tetsim.h (format and validation live in tetsim_amd/csrc/mesh_file.cpp; nothing is re-implemented here).

    write_mesh("dragon.tetsim", verts, tets, edge_ids=..., vis_verts=..., vis_tri_ids=..., tet_colour=..., vert_owner=..., part_count=8)
    with MeshFile("dragon.tetsim") as m:  m.verts, m.tets, m.tet_colour ...      # numpy views of the mapping
    body = SoftBodyHIP.fromFile("dragon.tetsim", physicsParams, solver="neohookean", order="coloured")

CLI:  python -m tetsim_amd.meshfile pack  -o out.tetsim --verts v.f32 --tets t.i32 [--vis vis.f32] [--colour] [--parts N]
      python -m tetsim_amd.meshfile info  file.tetsim
"""
import ctypes as C

import numpy as np

from . import _capi as capi


def _arr(a, dtype, cols):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=dtype)
    if a.size % cols:
        raise ValueError("array length is not a multiple of %d" % cols)
    return a


def write_mesh(path, verts, tets, edge_ids=None, vis_verts=None, vis_tri_ids=None, tet_colour=None, vert_owner=None, part_count=0):
    """Write the reference's five mesh arrays (+ optional colouring / partition map) as one .tetsim file."""
    v, t = _arr(verts, np.float32, 3), _arr(tets, np.int32, 4)
    e, vv, vt = _arr(edge_ids, np.int32, 2), _arr(vis_verts, np.float32, 4), _arr(vis_tri_ids, np.int32, 3)
    col, own = _arr(tet_colour, np.int32, 1), _arr(vert_owner, np.int32, 1)
    if col is not None and col.size != t.size // 4:
        raise ValueError("tet_colour needs one entry per tet")
    if own is not None and own.size != v.size // 3:
        raise ValueError("vert_owner needs one entry per particle")
    a = capi.TetSimMeshArrays()
    a.num_particles, a.num_elems = v.size // 3, t.size // 4
    a.verts = v.ctypes.data_as(C.POINTER(C.c_float))
    a.tets = t.ctypes.data_as(C.POINTER(C.c_int32))
    for arr, cnt, ptr, cols, ty in ((e, "num_edges", "edge_ids", 2, C.c_int32), (vv, "num_vis_verts", "vis_verts", 4, C.c_float),
                                    (vt, "num_vis_tris", "vis_tri_ids", 3, C.c_int32)):
        if arr is not None:
            setattr(a, cnt, arr.size // cols)
            setattr(a, ptr, arr.ctypes.data_as(C.POINTER(ty)))
    if col is not None:
        a.tet_colour = col.ctypes.data_as(C.POINTER(C.c_int32))
    if own is not None:
        a.vert_owner = own.ctypes.data_as(C.POINTER(C.c_int32))
        a.part_count = int(part_count) if part_count else int(own.max()) + 1
    capi.check(capi.lib().tetsim_mesh_write(str(path).encode(), C.byref(a)))


class MeshFile:
    """Read-only view of a .tetsim file: numpy arrays that alias the library's mmap (valid until close())."""

    def __init__(self, path):
        self._L = capi.lib()
        self._h = C.c_void_p()
        capi.check(self._L.tetsim_mesh_open(str(path).encode(), C.byref(self._h)))
        a = capi.TetSimMeshArrays()
        capi.check(self._L.tetsim_mesh_arrays(self._h, C.byref(a)))
        self.num_particles, self.num_elems, self.part_count = a.num_particles, a.num_elems, a.part_count

        def view(ptr, rows, cols):
            if not ptr:
                return None
            if rows == 0:
                return np.empty((0, cols) if cols > 1 else (0,), dtype=np.float32 if ptr._type_ is C.c_float else np.int32)
            out = np.ctypeslib.as_array(ptr, shape=(rows * cols,))
            return out.reshape(rows, cols) if cols > 1 else out
        self.verts = view(a.verts, a.num_particles, 3)
        self.tets = view(a.tets, a.num_elems, 4)
        self.edge_ids = view(a.edge_ids, a.num_edges, 2)
        self.vis_verts = view(a.vis_verts, a.num_vis_verts, 4)
        self.vis_tri_ids = view(a.vis_tri_ids, a.num_vis_tris, 3)
        self.tet_colour = view(a.tet_colour, a.num_elems, 1)
        self.vert_owner = view(a.vert_owner, a.num_particles, 1)

    def close(self):
        if self._h:
            for k in ("verts", "tets", "edge_ids", "vis_verts", "vis_tri_ids", "tet_colour", "vert_owner"):
                setattr(self, k, None)   # the views die with the mapping
            self._L.tetsim_mesh_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def greedy_colours(tets, num_particles):
    """The library's greedy colouring (tetsim_prep_colours) as an array, e.g. to store it in a file."""
    t = _arr(tets, np.int32, 4)
    out = np.empty(t.size // 4, dtype=np.int32)
    n = C.c_uint32()
    capi.check(capi.lib().tetsim_prep_colours(t.ctypes.data_as(C.POINTER(C.c_int32)), t.size // 4, int(num_particles),
                                              out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
    return out, int(n.value)


def _main(argv):
    import argparse
    ap = argparse.ArgumentParser(prog="python -m tetsim_amd.meshfile")
    sub = ap.add_subparsers(dest="cmd", required=True)
    pk = sub.add_parser("pack", help="raw little-endian arrays -> .tetsim")
    pk.add_argument("-o", "--out", required=True)
    pk.add_argument("--verts", required=True, help="f32 xyz per particle")
    pk.add_argument("--tets", required=True, help="i32, 4 ids per tet")
    pk.add_argument("--edges", help="i32, 2 ids per edge")
    pk.add_argument("--vis", help="f32 [tetNr,b0,b1,b2] per embedded vertex")
    pk.add_argument("--vis-tris", help="i32, 3 ids per triangle")
    pk.add_argument("--colour", action="store_true", help="store the greedy tet colouring")
    pk.add_argument("--parts", type=int, default=0, help="store the built-in partitioner's particle -> partition map for N parts (tetsim_prep_partition)")
    inf = sub.add_parser("info", help="print the sections of a .tetsim file")
    inf.add_argument("file")
    args = ap.parse_args(argv)
    if args.cmd == "pack":
        v = np.fromfile(args.verts, dtype="<f4")
        t = np.fromfile(args.tets, dtype="<i4")
        kw = {}
        if args.edges:
            kw["edge_ids"] = np.fromfile(args.edges, dtype="<i4")
        if args.vis:
            kw["vis_verts"] = np.fromfile(args.vis, dtype="<f4")
        if args.vis_tris:
            kw["vis_tri_ids"] = np.fromfile(args.vis_tris, dtype="<i4")
        if args.colour:
            kw["tet_colour"], n = greedy_colours(t, v.size // 3)
            print("colours:", n)
        if args.parts > 1:
            from .partition import partition, partition_quality   # the built-in partitioner, with the coordinates' candidates
            kw["vert_owner"] = partition(t, v.size // 3, args.parts, v)
            kw["part_count"] = args.parts
            q = partition_quality(t, v.size // 3, args.parts, kw["vert_owner"])
            print("partition: %d parts, ghost particles %.1f%%, ghost tets %.1f%%, local-tet imbalance %.1f%%" %
                  (args.parts, 100 * q["ghost_particle_fraction"], 100 * q["ghost_tet_fraction"], 100 * q["local_tet_imbalance"]))
        write_mesh(args.out, v, t, **kw)
        print("wrote", args.out)
    else:
        with MeshFile(args.file) as m:
            print("particles %d  tets %d  part_count %d" % (m.num_particles, m.num_elems, m.part_count))
            for k in ("edge_ids", "vis_verts", "vis_tri_ids", "tet_colour", "vert_owner"):
                a = getattr(m, k)
                print("  %-12s %s" % (k, "absent" if a is None else str(a.shape)))


if __name__ == "__main__":
    import sys
    _main(sys.argv[1:])
