"""ctypes loader for oracle/libtetsim_oracle.so (TEST INFRASTRUCTURE ONLY).

OracleNH restates /root/reference/src/Softbody.js (Neo-Hookean XPBD, Gauss-Seidel; f64 arithmetic,
f32 stores).  OraclePJ restates the GLSL passes of /root/reference/src/SoftbodyGPU.js:59-376
(shape-matching polar-decomposition Jacobi; f32).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtetsim_oracle.so")
_lib = None


class OrcParams(C.Structure):
    _fields_ = [("gravity", C.c_double), ("friction", C.c_double), ("devCompliance", C.c_double),
                ("volCompliance", C.c_double), ("worldBounds", C.c_double * 6)]

    @classmethod
    def from_dict(cls, d):
        p = cls()
        p.gravity = d.get("gravity", -9.81)
        p.friction = d.get("friction", 1000.0)
        p.devCompliance = d.get("devCompliance", 1.0 / 100000.0)
        p.volCompliance = d.get("volCompliance", 0.0)
        wb = d.get("worldBounds", [-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
        for i in range(6):
            p.worldBounds[i] = wb[i]
        return p


def build_oracle(force=False):
    src = os.path.join(_HERE, "tetsim_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libtetsim_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build_oracle()
    lib = C.CDLL(_SO)
    fp, ip, vp = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p
    lib.orc_nh_create.restype = vp
    lib.orc_nh_create.argtypes = [fp, C.c_int, ip, C.c_int, C.c_double]
    lib.orc_nh_destroy.argtypes = [vp]
    lib.orc_nh_simulate.argtypes = [vp, C.c_double, C.POINTER(OrcParams)]
    lib.orc_nh_set_grab.argtypes = [vp, C.c_int, fp]
    lib.orc_nh_start_grab.restype = C.c_int
    lib.orc_nh_start_grab.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    for n in ("pos", "prev", "vel", "inv_mass", "inv_rest_pose", "inv_rest_volume"):
        f = getattr(lib, "orc_nh_" + n)
        f.restype, f.argtypes = fp, [vp]
    lib.orc_nh_vol_error.restype = C.c_double
    lib.orc_nh_vol_error.argtypes = [vp]
    lib.orc_pj_create.restype = vp
    lib.orc_pj_create.argtypes = [fp, C.c_int, ip, C.c_int, C.c_double, C.c_int]
    lib.orc_pj_destroy.argtypes = [vp]
    lib.orc_pj_simulate.argtypes = [vp, C.c_double, C.POINTER(OrcParams)]
    lib.orc_pj_set_grab.argtypes = [vp, C.c_int, fp]
    lib.orc_pj_set_ref_grab_texel.argtypes = [vp, C.c_int]
    for n in ("pos", "prev", "vel", "quat"):
        getattr(lib, "orc_pj_read_" + n).argtypes = [vp, fp]
    lib.orc_pj_read_elem.argtypes = [vp, C.c_int, fp]
    lib.orc_pj_write_particles.argtypes = [vp, C.c_int, ip, fp, fp]
    lib.orc_pj_write_tets.argtypes = [vp, C.c_int, ip, fp, fp]
    lib.orc_pj_slots.restype, lib.orc_pj_slots.argtypes = ip, [vp]
    lib.orc_pj_inv_rest_volume.restype, lib.orc_pj_inv_rest_volume.argtypes = fp, [vp]
    lib.orc_pj_inv_mass.restype, lib.orc_pj_inv_mass.argtypes = fp, [vp]
    lib.orc_pj_biggest_table.restype, lib.orc_pj_biggest_table.argtypes = C.c_int, [vp]
    lib.orc_pj_iter_hist.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.orc_max_threads.restype = C.c_int
    lib.orc_set_threads.argtypes = [C.c_int]
    lib.orc_set_threads(1)  # deterministic and fastest for the small test meshes; bench.py opts into more
    _lib = lib
    return lib


def max_threads():
    return _load().orc_max_threads()


def set_threads(n):
    _load().orc_set_threads(int(n))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _Base:
    def __init__(self, verts, tets):
        self._verts = _f32(verts).reshape(-1, 3)
        self._tets = np.ascontiguousarray(tets, dtype=np.int32).reshape(-1, 4)
        self.numParticles = self._verts.shape[0]
        self.numElems = self._tets.shape[0]
        self.grabId = -1

    def _params(self, physicsParams):
        return physicsParams if isinstance(physicsParams, OrcParams) else OrcParams.from_dict(physicsParams)


class OracleNH(_Base):
    """Restatement of the reference `SoftBody` (Softbody.js:3-412) -- physics only."""

    def __init__(self, vertices, tetIds, physicsParams):
        super().__init__(vertices, tetIds)
        self._lib = _load()
        density = physicsParams.get("density", 1000.0) if isinstance(physicsParams, dict) else 1000.0
        self._h = self._lib.orc_nh_create(_fptr(self._verts), self.numParticles,
                                          self._tets.ctypes.data_as(C.POINTER(C.c_int32)), self.numElems, density)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.orc_nh_destroy(self._h)
            self._h = None

    def simulate(self, dt, physicsParams):
        p = self._params(physicsParams)
        self._lib.orc_nh_simulate(self._h, float(dt), C.byref(p))

    def _arr(self, name, n):
        ptr = getattr(self._lib, "orc_nh_" + name)(self._h)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    @property
    def pos(self):
        return self._arr("pos", 3 * self.numParticles).reshape(-1, 3)

    @property
    def prevPos(self):
        return self._arr("prev", 3 * self.numParticles).reshape(-1, 3)

    @property
    def vel(self):
        return self._arr("vel", 3 * self.numParticles).reshape(-1, 3)

    @property
    def invMass(self):
        return self._arr("inv_mass", self.numParticles)

    @property
    def invRestPose(self):
        return self._arr("inv_rest_pose", 9 * self.numElems)

    @property
    def invRestVolume(self):
        return self._arr("inv_rest_volume", self.numElems)

    @property
    def volError(self):
        return self._lib.orc_nh_vol_error(self._h)

    def startGrab(self, x, y, z):
        self.grabId = self._lib.orc_nh_start_grab(self._h, x, y, z)
        return self.grabId

    def setGrab(self, gid, xyz=None):
        self.grabId = gid
        a = _f32(xyz if xyz is not None else [0, 0, 0])
        self._lib.orc_nh_set_grab(self._h, int(gid), _fptr(a))

    def moveGrabbed(self, x, y, z):
        self.setGrab(self.grabId, [x, y, z])

    def endGrab(self):
        self.setGrab(-1)


class OraclePJ(_Base):
    """Restatement of the reference `SoftBodyGPU` GLSL passes (SoftbodyGPU.js:59-376) -- physics only."""

    def __init__(self, vertices, tetIds, physicsParams, slot_quirk=True, ref_grab_texel=False):
        super().__init__(vertices, tetIds)
        self._lib = _load()
        density = physicsParams.get("density", 1000.0) if isinstance(physicsParams, dict) else 1000.0
        self._h = self._lib.orc_pj_create(_fptr(self._verts), self.numParticles,
                                          self._tets.ctypes.data_as(C.POINTER(C.c_int32)), self.numElems,
                                          density, 1 if slot_quirk else 0)
        # the reference's indexFromUV quirk (SoftbodyGPU.js:335-338): only for pinning against its golden vectors
        self._lib.orc_pj_set_ref_grab_texel(self._h, 1 if ref_grab_texel else 0)

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.orc_pj_destroy(self._h)
            self._h = None

    def simulate(self, dt, physicsParams):
        p = self._params(physicsParams)
        self._lib.orc_pj_simulate(self._h, float(dt), C.byref(p))

    def _read(self, name, n, cols):
        out = np.empty(n * cols, dtype=np.float32)
        getattr(self._lib, "orc_pj_read_" + name)(self._h, _fptr(out))
        return out.reshape(n, cols)

    @property
    def pos(self):
        return self._read("pos", self.numParticles, 3)

    @property
    def prevPos(self):
        return self._read("prev", self.numParticles, 3)

    @property
    def vel(self):
        return self._read("vel", self.numParticles, 3)

    @property
    def quats(self):
        return self._read("quat", self.numElems, 4)

    def elems(self, k):
        out = np.empty(self.numElems * 4, dtype=np.float32)
        self._lib.orc_pj_read_elem(self._h, k, _fptr(out))
        return out.reshape(-1, 4)

    @property
    def slots(self):
        p = self._lib.orc_pj_slots(self._h)
        return np.ctypeslib.as_array(p, shape=(self.numParticles, 36)).copy()

    @property
    def invRestVolume(self):
        return np.ctypeslib.as_array(self._lib.orc_pj_inv_rest_volume(self._h), shape=(self.numElems,)).copy()

    @property
    def invMass(self):
        return np.ctypeslib.as_array(self._lib.orc_pj_inv_mass(self._h), shape=(self.numParticles,)).copy()

    @property
    def biggestT(self):
        return self._lib.orc_pj_biggest_table(self._h)

    @property
    def iterHist(self):
        h = (C.c_longlong * 10)()
        self._lib.orc_pj_iter_hist(self._h, h)
        return list(h)

    def setGrab(self, gid, xyz=None):
        self.grabId = gid
        a = _f32(xyz if xyz is not None else [0, 0, 0])
        self._lib.orc_pj_set_grab(self._h, int(gid), _fptr(a))

    def endGrab(self):
        self.setGrab(-1)

    def tetState(self, idx):
        """(quaternions [n,4], carried rest corners [n,4,4]) of the listed tets."""
        idx = np.asarray(idx, dtype=np.int64)
        q = self.quats[idx]
        el = np.stack([self.elems(k)[idx] for k in range(4)], axis=1)
        return q, el

    def writeTets(self, idx, quats, elems):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        q, el = _f32(quats).reshape(-1), _f32(elems).reshape(-1)
        self._lib.orc_pj_write_tets(self._h, len(idx), idx.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(q), _fptr(el))

    def writeParticles(self, idx, pos, vel):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        p, v = _f32(pos).reshape(-1), _f32(vel).reshape(-1)
        self._lib.orc_pj_write_particles(self._h, len(idx), idx.ctypes.data_as(C.POINTER(C.c_int32)), _fptr(p), _fptr(v))
