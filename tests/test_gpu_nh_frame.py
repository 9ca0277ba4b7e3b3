"""Small Neo-Hookean bodies (BASELINE config 1: the Dragon through the reference's CPU algorithm): tetsim_step_n and tetsim_step run a
whole call as ONE single-workgroup launch with every particle in LDS (nh_kernels.inc: nh_frame_kernel, TetSimInfo.fused_particle_pass
== 4).  Its contract: a call of n substeps equals n calls of one substep and equals the level kernels (one launch per level, what
tetsim_profile steps with) BIT FOR BIT in both arithmetics -- positions, velocities, previous positions, volError -- and therefore
the reference's Softbody.js where the level kernels do (tests/test_gpu_neohookean.py runs its golden hashes through this kernel)."""
import numpy as np
import pytest

from conftest import load_mesh
from tetsim_amd import SoftBodyHIP, make_lattice

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 10


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("precision", ["precise", "fast"])
@pytest.mark.parametrize("order", ["original", "coloured"])
def test_one_launch_per_call_equals_one_launch_per_level(order, precision):
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.003, 0.0])      # 3 mm above the floor: contact and friction from the second frame on
    a = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order=order, precision=precision)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order=order, precision=precision)
    c = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order=order, precision=precision)
    assert a.info.fused_particle_pass == 4
    for k, n in enumerate((10, 1, 3, 10, 7)):
        if k == 1:
            for body in (a, b, c):
                body.setGrab(100, [0.1, 0.9, -0.1])
        if k == 3:
            for body in (a, b, c):
                body.endGrab()
        a.simulateSubsteps(n, DT, PP)          # one launch for the n substeps
        for _ in range(n):
            b.simulate(DT, PP)                 # one launch per substep
        pr = c.profile(n, DT, PP)              # the level kernels: one launch per level, a prediction and a particle pass per substep
        assert pr["tet_launches"] == n * c.info.num_levels
        for other in (b, c):
            assert _same(a.pos, other.pos) and _same(a.vel, other.vel) and _same(a.prevPos, other.prevPos), (k, n)
            assert a.volError == other.volError, (k, n)
    assert np.isfinite(a.pos).all() and a.pos[:, 1].min() == 0.0


def test_which_bodies_take_it():
    v, t = load_mesh("dragon")
    assert SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="clustered", precision="fast").info.fused_particle_pass == 0
    v, t = make_lattice(15)      # 16^3 = 4,096 particles x 40 B = all 160 KB of a CU's LDS
    a = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured", precision="fast")
    b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured", precision="fast")
    assert a.info.fused_particle_pass == 4
    a.simulateSubsteps(12, DT, PP)
    for _ in range(12):
        b.simulate(DT, PP)
    assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and a.volError == b.volError
    # PRECISE stops at 12,288 tets (f64 on one CU is throughput-bound beyond): the same 20,250-tet body keeps one launch per level
    assert SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured", precision="precise").info.fused_particle_pass == 0
    v, t = make_lattice(16)      # 4,913 particles: one launch per level
    assert SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured", precision="fast").info.fused_particle_pass == 0
