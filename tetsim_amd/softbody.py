"""Host-side mirror of the reference's solver objects over the C ABI (include/tetsim.h).

`SoftBodyHIP` keeps the constructor + simulate()/endFrame()/grab surface of the reference's
`SoftBody` (/root/reference/src/Softbody.js:3-298) and `SoftBodyGPU` (SoftbodyGPU.js:4-712) so the
caller's loop (main.js:79-89) is unchanged:

    body = SoftBodyHIP(vertices, tetIds, tetEdgeIds, physicsParams, visVerts, visTriIds, visMaterial, world)
    for step in range(numSubsteps): body.simulate(dt, physicsParams)
    body.endFrame()

Only the physics is here (display meshes are three.js objects of the JavaScript host; the N-API twin of this
class, tetsim_amd/node/SoftBodyHIP.js, owns those).  All compute happens in libtetsim_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import TetSimError  # noqa: F401


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def make_params(physicsParams):
    """physicsParams object (main.js:22-36) -> TetSimParams."""
    p = capi.TetSimParams()
    capi.lib().tetsim_default_params(C.byref(p))
    g = physicsParams.get if isinstance(physicsParams, dict) else (lambda k, d=None: getattr(physicsParams, k, d))
    for k in ("gravity", "friction", "devCompliance", "volCompliance"):
        v = g(k, None)
        if v is not None:
            setattr(p, k, float(v))
    wb = g("worldBounds", None)
    if wb is not None:
        for i in range(6):
            p.worldBounds[i] = float(wb[i])
    return p


class SoftBodyHIP:
    """Drop-in for `new SoftBody(...)` / `new SoftBodyGPU(...)`; `solver` picks which one is mirrored.

    solver="polar"  -> SoftBodyGPU's shape-matching Jacobi (default, like the reference's GPU path)
    solver="neohookean" -> SoftBody's Neo-Hookean XPBD Gauss-Seidel
    """

    def __init__(self, vertices, tetIds, tetEdgeIds=None, physicsParams=None, visVerts=None, visTriIds=None,
                 visMaterial=None, world=None, *, solver="polar", precision="precise", order="original",
                 ref_slot_table=True, ref_fixed_bounds=True, gather=False, constant_rest_shape=False, ref_grab_texel=False, deep_ghosts=False, ref_rotation_exit=False,
                 lean_state=False,
                 device=0, part_count=1, part_index=0, vert_owner=None, tet_colour=None, mesh_file=None, batch=None):
        L = capi.lib()
        self.physicsParams = physicsParams if physicsParams is not None else {}
        self._batch = None
        if batch is not None:   # SoftBodyHIP.batch: several independent bodies behind one handle (tetsim_create_batch)
            self._batch = [(_f32(v).reshape(-1), np.ascontiguousarray(np.asarray(t).reshape(-1), dtype=np.int32)) for v, t in batch]
            offs = np.concatenate([[0], np.cumsum([len(v) // 3 for v, _ in self._batch])])
            vertices = np.concatenate([v for v, _ in self._batch])
            tetIds = np.concatenate([t + int(offs[i]) for i, (_, t) in enumerate(self._batch)])
        if mesh_file is not None:   # SoftBodyHIP.fromFile: arrays come from the .tetsim container
            from .meshfile import MeshFile
            with MeshFile(mesh_file) as mf:
                vertices, tetIds = mf.verts.copy(), mf.tets.copy()
                if visVerts is None and mf.vis_verts is not None:
                    visVerts = mf.vis_verts.copy()
                    if visTriIds is None and mf.vis_tri_ids is not None:
                        visTriIds = mf.vis_tri_ids.copy()
        self._verts = _f32(vertices).reshape(-1)
        self._tets = np.ascontiguousarray(np.asarray(tetIds).reshape(-1), dtype=np.int32)
        if self._verts.size % 3 or self._tets.size % 4:
            raise ValueError("vertices must hold 3 floats per particle and tetIds 4 ids per tet")
        self.numParticles = self._verts.size // 3   # Softbody.js:9
        self.numElems = self._tets.size // 4        # Softbody.js:10
        self.visVerts = visVerts
        self.grabId = -1
        self.grabPos = np.zeros(3, dtype=np.float32)
        o = capi.TetSimOptions()
        L.tetsim_default_options(C.byref(o))
        o.solver = {"polar": capi.SOLVER_POLAR_JACOBI, "neohookean": capi.SOLVER_NEOHOOKEAN_GS}[solver]
        o.precision = {"precise": capi.PRECISE, "fast": capi.FAST}[precision]
        o.order = {"original": capi.ORDER_ORIGINAL, "coloured": capi.ORDER_COLOURED, "clustered": capi.ORDER_CLUSTERED}[order]
        o.flags = ((capi.FLAG_REF_SLOT_TABLE if ref_slot_table else 0) | (capi.FLAG_REF_FIXED_BOUNDS if ref_fixed_bounds else 0)
                   | (capi.FLAG_GATHER_FORMULATION if gather else 0)
                   | (capi.FLAG_CONSTANT_REST_SHAPE if constant_rest_shape else 0)
                   | (capi.FLAG_REF_GRAB_TEXEL if ref_grab_texel else 0) | (capi.FLAG_DEEP_GHOSTS if deep_ghosts else 0)
                   | (capi.FLAG_REF_ROTATION_EXIT if ref_rotation_exit else 0) | (capi.FLAG_LEAN_STATE if lean_state else 0))
        o.device = device
        d = self.physicsParams.get("density", 1000.0) if isinstance(self.physicsParams, dict) else getattr(self.physicsParams, "density", 1000.0)
        o.density = float(d)
        o.part_count, o.part_index = part_count, part_index
        self._owner = None
        if vert_owner is not None:
            self._owner = np.ascontiguousarray(vert_owner, dtype=np.int32)
            o.vert_owner = _ip(self._owner)
        self._colour = None
        if tet_colour is not None:
            self._colour = np.ascontiguousarray(tet_colour, dtype=np.int32)
            if self._colour.size != self.numElems:
                raise ValueError("tet_colour needs one entry per tet")
            o.tet_colour = _ip(self._colour)
        self.solver = solver
        self._h = C.c_void_p()
        if mesh_file is not None:   # the library maps the file itself and picks up a stored colouring / partition map
            capi.check(L.tetsim_create_from_file(str(mesh_file).encode(), C.byref(o), C.byref(self._h)))
        elif self._batch is not None:
            n = len(self._batch)
            vp = (C.POINTER(C.c_float) * n)(*[_fp(v) for v, _ in self._batch])
            tp = (C.POINTER(C.c_int32) * n)(*[_ip(t) for _, t in self._batch])
            nvs = (C.c_uint32 * n)(*[len(v) // 3 for v, _ in self._batch])
            nts = (C.c_uint32 * n)(*[len(t) // 4 for _, t in self._batch])
            capi.check(L.tetsim_create_batch(vp, nvs, tp, nts, n, C.byref(o), C.byref(self._h)))
        else:
            capi.check(L.tetsim_create(_fp(self._verts), self.numParticles, _ip(self._tets), self.numElems,
                                       C.byref(o), C.byref(self._h)))
        self.info = capi.TetSimInfo()
        capi.check(L.tetsim_get_info(self._h, C.byref(self.info)), self._h)
        self._L = L
        self.numVisVerts = 0
        if visVerts is not None and len(visVerts):   # Softbody.js:46-47: rows (tetNr, b0, b1, b2); a partition keeps the rows of the tets it owns (visualIds)
            if mesh_file is not None:   # tetsim_create_from_file attached the stored visual mesh already
                self.numVisVerts, self._has_normals = self.info.num_vis_verts, False
            else:
                self.setVisualMesh(visVerts)
            if visTriIds is not None and len(visTriIds) and part_count <= 1:   # Softbody.js:48-50: enables visualVertexNormals() (unpartitioned bodies)
                self.setVisualTriangles(visTriIds)

    @classmethod
    def batch(cls, bodies, physicsParams=None, **kw):
        """Several INDEPENDENT bodies `[(vertices, tetIds), ...]` behind one handle: one launch per kernel steps them all (the
        reference steps softBodies[] one after the other, main.js:80-84).  `.pos` etc. are the concatenation; `bodyRanges` gives
        each body's particle / tet range.  Every body's results equal its solo run bit for bit."""
        return cls(None, None, None, physicsParams, batch=list(bodies), **kw)

    @property
    def bodyRanges(self):
        n = self.info.num_bodies
        fp_, fe = (C.c_uint32 * (n + 1))(), (C.c_uint32 * (n + 1))()
        capi.check(self._L.tetsim_get_batch_layout(self._h, fp_, fe), self._h)
        return [((fp_[b], fp_[b + 1]), (fe[b], fe[b + 1])) for b in range(n)]

    @classmethod
    def fromFile(cls, path, physicsParams=None, visMaterial=None, world=None, **kw):
        """Build the body from a .tetsim container (tetsim_amd/meshfile.py) instead of the five Dragon.js arrays."""
        return cls(None, None, None, physicsParams, None, None, visMaterial, world, mesh_file=path, **kw)

    # -- lifecycle ------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.tetsim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the hot path ---------------------------------------------------------------------------------
    def simulate(self, dt, physicsParams=None):
        """One substep (Softbody.js:195 / SoftbodyGPU.js:610).  Asynchronous."""
        pp = self.physicsParams if physicsParams is None else physicsParams
        if isinstance(pp, dict) and self.solver == "polar":
            pp["dt"] = dt  # SoftbodyGPU.js:611 writes dt back into the caller's object
        capi.check(self._L.tetsim_step(self._h, float(dt), C.byref(make_params(pp))), self._h)

    def simulateSubsteps(self, n, dt, physicsParams=None):
        """The caller's whole substep loop (main.js:79-84) as ONE FFI crossing / one HIP-graph launch."""
        pp = self.physicsParams if physicsParams is None else physicsParams
        capi.check(self._L.tetsim_step_n(self._h, int(n), float(dt), C.byref(make_params(pp))), self._h)

    def endFrame(self):
        """Softbody.js:244-247 refreshes the display meshes from .pos; here: make the frame's results visible."""
        capi.check(self._L.tetsim_sync(self._h), self._h)

    def sync(self):
        capi.check(self._L.tetsim_sync(self._h), self._h)

    # -- state ----------------------------------------------------------------------------------------
    def _read3(self, fn, n):
        out = np.empty(3 * n, dtype=np.float32)
        capi.check(fn(self._h, _fp(out)), self._h)
        return out.reshape(-1, 3)

    @property
    def pos(self):
        return self._read3(self._L.tetsim_read_positions, self.info.owned_particles)

    @property
    def posPinned(self):
        """Zero-copy read-back: a numpy VIEW of the handle's pinned host buffer (valid until close(), refreshed by every
        access): one device pack kernel + one DMA, no intermediate host copy."""
        ptr = C.POINTER(C.c_float)()
        capi.check(self._L.tetsim_read_positions_pinned(self._h, C.byref(ptr)), self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.info.owned_particles, 3))

    @property
    def prevPos(self):
        return self._read3(self._L.tetsim_read_prev_positions, self.info.owned_particles)

    @property
    def vel(self):
        return self._read3(self._L.tetsim_read_velocities, self.info.owned_particles)

    @property
    def quats(self):
        out = np.empty(4 * self.info.local_elems, dtype=np.float32)
        capi.check(self._L.tetsim_read_quats(self._h, _fp(out)), self._h)
        return out.reshape(-1, 4)

    @property
    def quatsPinned(self):
        """Zero-copy twin of `quats`: a numpy VIEW of the handle's pinned host buffer, refreshed by every access."""
        ptr = C.POINTER(C.c_float)()
        capi.check(self._L.tetsim_read_quats_pinned(self._h, C.byref(ptr)), self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.info.local_elems, 4))

    def saveState(self):
        """The complete solver state (positions, velocities, per-tet quaternions and carried rest shape) as bytes."""
        n = C.c_uint64()
        capi.check(self._L.tetsim_state_size(self._h, C.byref(n)), self._h)
        buf = (C.c_char * n.value)()
        capi.check(self._L.tetsim_save_state(self._h, buf, n.value), self._h)
        return bytes(buf.raw)

    def loadState(self, blob):
        """Resume from saveState() of a body with the same mesh and options: the trajectory continues bit for bit."""
        capi.check(self._L.tetsim_load_state(self._h, bytes(blob), len(blob)), self._h)

    @property
    def volError(self):
        v = C.c_double()
        capi.check(self._L.tetsim_read_vol_error(self._h, C.byref(v)), self._h)
        return v.value

    @property
    def invMass(self):
        out = np.empty(self.numParticles, dtype=np.float32)
        capi.check(self._L.tetsim_read_inv_mass(self._h, _fp(out)), self._h)
        return out

    @property
    def ownedIds(self):
        out = np.empty(self.info.owned_particles, dtype=np.int32)
        capi.check(self._L.tetsim_get_owned_ids(self._h, _ip(out)), self._h)
        return out

    @property
    def localTets(self):
        out = np.empty(self.info.local_elems, dtype=np.int32)
        capi.check(self._L.tetsim_get_local_tets(self._h, _ip(out)), self._h)
        return out

    @property
    def tetOrder(self):
        out = np.empty(self.numElems, dtype=np.int32)
        capi.check(self._L.tetsim_get_tet_order(self._h, _ip(out)), self._h)
        return out

    @property
    def levelOffsets(self):
        out = np.empty(self.info.num_levels + 1, dtype=np.int32)
        capi.check(self._L.tetsim_get_level_offsets(self._h, _ip(out)), self._h)
        return out

    def writeState(self, pos, vel):
        p, v = _f32(pos).reshape(-1), _f32(vel).reshape(-1)
        capi.check(self._L.tetsim_write_state(self._h, _fp(p), _fp(v)), self._h)

    # -- embedded visual mesh (Softbody.js:259-277 / SoftbodyGPU.js:424-448), skinned on the device ---------------
    def setVisualMesh(self, visVerts, restNormals=None):
        vv = _f32(visVerts).reshape(-1)
        n0 = None if restNormals is None else _f32(restNormals).reshape(-1)
        capi.check(self._L.tetsim_set_visual_mesh(self._h, _fp(vv), vv.size // 4, _fp(n0) if n0 is not None else None), self._h)
        self._has_normals = n0 is not None
        capi.check(self._L.tetsim_get_info(self._h, C.byref(self.info)), self._h)
        self.numVisVerts = self.info.num_vis_verts   # (a partition keeps the rows of the tets it owns: visualIds)

    @property
    def visualIds(self):
        """Row of the caller's visVerts behind each attached visual vertex (identity when unpartitioned; a partition: its own rows)."""
        out = np.empty(self.numVisVerts, dtype=np.int32)
        capi.check(self._L.tetsim_get_visual_ids(self._h, _ip(out)), self._h)
        return out

    def refreshFinalGhosts(self):
        """RCCL partitions, every rank together, after the frame's last substep and before visualPositions(): the ghost particles'
        end-of-substep positions from their owners (in-process groups: group_refresh_final).  The read itself never communicates."""
        capi.check(self._L.tetsim_halo_refresh_final(self._h), self._h)

    def visualPositions(self, with_normals=False):
        out = np.empty(3 * self.numVisVerts, dtype=np.float32)
        nrm = np.empty(3 * self.numVisVerts, dtype=np.float32) if with_normals else None
        capi.check(self._L.tetsim_read_visual_mesh(self._h, _fp(out), _fp(nrm) if with_normals else None), self._h)
        return (out.reshape(-1, 3), nrm.reshape(-1, 3)) if with_normals else out.reshape(-1, 3)

    def setVisualTriangles(self, visTriIds):
        """The visual mesh's triangle list (the reference's `visTriIds`, Softbody.js:48-50): enables visualVertexNormals()."""
        tri = np.ascontiguousarray(np.asarray(visTriIds).reshape(-1), dtype=np.int32)
        capi.check(self._L.tetsim_set_visual_triangles(self._h, tri.ctypes.data_as(C.POINTER(C.c_int32)), tri.size // 3), self._h)

    def visualVertexNormals(self):
        """`visMesh.geometry.computeVertexNormals()` (Softbody.js:273) evaluated on the device, bit-exact with three.js r160."""
        out = np.empty(3 * self.numVisVerts, dtype=np.float32)
        capi.check(self._L.tetsim_read_visual_vertex_normals(self._h, _fp(out)), self._h)
        return out.reshape(-1, 3)

    def visualVertexNormalsFrom(self, allPositions):
        """computeVertexNormals of THIS body's rows from a complete set of visual positions [rows of visVerts, 3] -- a partition: the
        ranks' skins put together (visualPositions() scattered by visualIds); include/tetsim.h: tetsim_visual_vertex_normals_from."""
        ap = _f32(allPositions).reshape(-1)
        out = np.empty(3 * self.numVisVerts, dtype=np.float32)
        capi.check(self._L.tetsim_visual_vertex_normals_from(self._h, _fp(ap), _fp(out)), self._h)
        return out.reshape(-1, 3)

    # -- caller-provided transports (include/tetsim.h: tetsim_get_halo_plan / tetsim_halo_export / tetsim_halo_import) --
    def haloPlan(self):
        """[(neighbour rank, global ids sent, global ids received)] in neighbour-slot order."""
        n = self.info.num_neighbours
        if n == 0:
            return []
        neigh, sc, rc = (np.empty(n, dtype=np.int32) for _ in range(3))
        capi.check(self._L.tetsim_get_halo_plan(self._h, _ip(neigh), _ip(sc), _ip(rc), None, None), self._h)
        sid, rid = np.empty(int(sc.sum()), dtype=np.int32), np.empty(int(rc.sum()), dtype=np.int32)
        capi.check(self._L.tetsim_get_halo_plan(self._h, _ip(neigh), _ip(sc), _ip(rc), _ip(sid), _ip(rid)), self._h)
        so, ro = np.concatenate([[0], np.cumsum(sc)]), np.concatenate([[0], np.cumsum(rc)])
        return [(int(neigh[i]), sid[so[i]:so[i + 1]].copy(), rid[ro[i]:ro[i + 1]].copy()) for i in range(n)]

    def haloExport(self, slot, count):
        """Predicted positions (x, y, z, w) this handle owes neighbour slot `slot`, as a host array [count, 4]."""
        out = np.empty(4 * count, dtype=np.float32)
        capi.check(self._L.tetsim_halo_export(self._h, int(slot), _fp(out)), self._h)
        return out.reshape(-1, 4)

    def haloImport(self, slot, xyzw):
        """Install the ghost predictions received from neighbour slot `slot`."""
        a = _f32(xyzw).reshape(-1)
        capi.check(self._L.tetsim_halo_import(self._h, int(slot), _fp(a)), self._h)

    # -- grab (Softbody.js:279-298) -------------------------------------------------------------------
    def startGrab(self, pos):
        p = _f32([pos["x"], pos["y"], pos["z"]] if isinstance(pos, dict) else pos)
        gid = C.c_int32(-1)
        capi.check(self._L.tetsim_start_grab(self._h, _fp(p), C.byref(gid)), self._h)
        self.grabId = gid.value
        self.grabPos[:] = p
        return self.grabId

    def nearestParticle(self, pos):
        """(global id, squared distance) of the owned particle nearest to pos -- the building block of startGrab for partitioned
        bodies: min over the partitions (ties: lowest id), then setGrab(id, pos) on each."""
        p = _f32([pos["x"], pos["y"], pos["z"]] if isinstance(pos, dict) else pos)
        gid, d2 = C.c_int32(), C.c_double()
        capi.check(self._L.tetsim_nearest_particle(self._h, _fp(p), C.byref(gid), C.byref(d2)), self._h)
        return int(gid.value), float(d2.value)

    def setGrab(self, gid, pos):
        p = _f32(pos)
        capi.check(self._L.tetsim_set_grab(self._h, int(gid), _fp(p)), self._h)
        self.grabId = int(gid)
        self.grabPos[:] = p

    def moveGrabbed(self, pos):
        p = _f32([pos["x"], pos["y"], pos["z"]] if isinstance(pos, dict) else pos)
        self.setGrab(self.grabId, p)

    def endGrab(self):
        capi.check(self._L.tetsim_set_grab(self._h, -1, None), self._h)
        self.grabId = -1

    # -- measurement ----------------------------------------------------------------------------------
    def profile(self, n, dt, physicsParams=None):
        pp = self.physicsParams if physicsParams is None else physicsParams
        pr = capi.TetSimProfile()
        capi.check(self._L.tetsim_profile(self._h, int(n), float(dt), C.byref(make_params(pp)), C.byref(pr)), self._h)
        return dict(total_ms=pr.total_ms, tet_ms=pr.kernel_ms[capi.K_TET], vertex_ms=pr.kernel_ms[capi.K_VERTEX],
                    tet_launches=pr.launches[capi.K_TET], vertex_launches=pr.launches[capi.K_VERTEX], substeps=pr.substeps,
                    tets_per_tet_launch=pr.tets_per_tet_launch)

    def timeKernels(self, reps, dt, physicsParams=None):
        """Kernel-only timing (back-to-back launches, one event pair per kernel class).  Scratch bodies only."""
        pp = self.physicsParams if physicsParams is None else physicsParams
        pr = capi.TetSimProfile()
        capi.check(self._L.tetsim_time_kernels(self._h, int(reps), float(dt), C.byref(make_params(pp)), C.byref(pr)), self._h)
        return dict(tet_us=pr.kernel_ms[capi.K_TET] / pr.launches[capi.K_TET] * 1e3,
                    vertex_us=pr.kernel_ms[capi.K_VERTEX] / pr.launches[capi.K_VERTEX] * 1e3,
                    tet_launches_per_substep=pr.launches[capi.K_TET] // reps)

    def timeSubsteps(self, n, dt, physicsParams=None):
        pp = self.physicsParams if physicsParams is None else physicsParams
        ms = C.c_double()
        capi.check(self._L.tetsim_time_step_n(self._h, int(n), float(dt), C.byref(make_params(pp)), C.byref(ms)), self._h)
        return ms.value


def halo_probe(body, reps=100):
    """{min, median, max} microseconds of one halo exchange of this rank with its real neighbours and message sizes (a collective of
    all ranks, between steps; include/tetsim.h: tetsim_halo_probe)."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    capi.check(capi.lib().tetsim_halo_probe(body._h, int(reps), C.byref(a), C.byref(b), C.byref(c)), body._h)
    return {"min": a.value, "median": b.value, "max": c.value}


def halo_p2p_probe(body, reps=100):
    """{min, median, max} microseconds of one hand-over of the peer-to-peer halo with all neighbours at once (a collective of all ranks,
    between steps; include/tetsim.h: tetsim_halo_p2p_probe)."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    capi.check(capi.lib().tetsim_halo_p2p_probe(body._h, int(reps), C.byref(a), C.byref(b), C.byref(c)), body._h)
    return {"min": a.value, "median": b.value, "max": c.value}


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it; the host distributes it to every rank)."""
    buf = (C.c_char * 128)()
    capi.check(capi.lib().tetsim_comm_unique_id(buf))
    return bytes(buf.raw)


def comm_init(body, uid, rank, nranks):
    capi.check(capi.lib().tetsim_comm_init(body._h, bytes(uid), int(rank), int(nranks)), body._h)


def comm_info(body):
    """What RCCL reports for this body's communicator + the halo volume of this partition (include/tetsim.h TetSimCommInfo)."""
    ci = capi.TetSimCommInfo()
    capi.check(capi.lib().tetsim_comm_info(body._h, C.byref(ci)), body._h)
    return {"rccl_ranks": ci.rccl_ranks, "rccl_rank": ci.rccl_rank, "neighbours": ci.neighbours,
            "send_bytes_per_substep": ci.send_bytes_per_substep, "recv_bytes_per_substep": ci.recv_bytes_per_substep,
            "max_message_bytes": ci.max_message_bytes, "loopback": bool(ci.loopback), "p2p": bool(ci.p2p)}


def comm_selftest(body):
    capi.check(capi.lib().tetsim_comm_selftest(body._h), body._h)


def measure_copy_bandwidth(nbytes, reps=20, device=0):
    g = C.c_double()
    capi.check(capi.lib().tetsim_measure_copy_bandwidth(device, int(nbytes), int(reps), C.byref(g)))
    return g.value


def measure_stream_bandwidth(nbytes, kind="copy", reps=20, device=0):
    """GB/s of the tuned streaming probe (include/tetsim.h: tetsim_measure_stream_bandwidth): kind copy (read + write bytes), read, write."""
    g = C.c_double()
    capi.check(capi.lib().tetsim_measure_stream_bandwidth(device, int(nbytes), int(reps), {"copy": 0, "read": 1, "write": 2}[kind], C.byref(g)))
    return g.value


def group_step_n(bodies, n, dt, physicsParams):
    """n substeps of every partition of one decomposition (same choreography as the RCCL path, in-process copies)."""
    arr = (C.c_void_p * len(bodies))(*[b._h for b in bodies])
    capi.check(capi.lib().tetsim_group_step_n(arr, len(bodies), int(n), float(dt), C.byref(make_params(physicsParams))))


def group_visual_vertex_normals(bodies, num_rows):
    """(positions, normals) of the WHOLE visual mesh [num_rows, 3] from the partitions of one process (after group_refresh_final):
    tetsim_group_read_visual_vertex_normals."""
    arr = (C.c_void_p * len(bodies))(*[b._h for b in bodies])
    pos, nrm = np.empty(3 * num_rows, dtype=np.float32), np.empty(3 * num_rows, dtype=np.float32)
    capi.check(capi.lib().tetsim_group_read_visual_vertex_normals(arr, len(bodies), _fp(pos), _fp(nrm)))
    return pos.reshape(-1, 3), nrm.reshape(-1, 3)


def group_refresh_final(bodies):
    """In-process group: every partition's ghost particles get their END-OF-SUBSTEP positions from their owners (what the visual mesh
    of a partition needs at the frame's end; the per-substep halo carries predictions)."""
    arr = (C.c_void_p * len(bodies))(*[b._h for b in bodies])
    capi.check(capi.lib().tetsim_group_refresh_final(arr, len(bodies)))


P2P_BLOB_BYTES = 512


def p2p_export(body):
    """This partition's buffers for the peer-to-peer halo (include/tetsim.h: tetsim_halo_p2p_export): TETSIM_P2P_BLOB_BYTES bytes
    that the host hands to every other rank."""
    buf = (C.c_char * P2P_BLOB_BYTES)()
    capi.check(capi.lib().tetsim_halo_p2p_export(body._h, buf), body._h)
    return bytes(buf.raw)


def p2p_connect(body, blobs):
    """blobs: one per partition, in rank order (a loopback body: its own, alone).  From the next step on the boundary-particle kernel
    stores the halo straight into the neighbours' ghost ranges; every rank must have connected before any of them steps."""
    raw = b"".join(bytes(b) for b in blobs)
    assert len(raw) == P2P_BLOB_BYTES * len(blobs)
    capi.check(capi.lib().tetsim_halo_p2p_connect(body._h, raw, len(blobs)), body._h)


def group_p2p_connect(bodies):
    """All partitions of one decomposition living in this process: switch their halo to peer-to-peer stores."""
    blobs = [p2p_export(b) for b in bodies]
    for b in bodies:
        p2p_connect(b, blobs)


def halo_exchange_local(bodies):
    arr = (C.c_void_p * len(bodies))(*[b._h for b in bodies])
    capi.check(capi.lib().tetsim_halo_exchange_local(arr, len(bodies)))
