"""The persistent frame kernel (pj_blocked.hip: pjb_frame_kernel): small unpartitioned FAST bodies run a whole
tetsim_step_n call as ONE launch -- every tile's workgroup resident for all n substeps, tet records and particles in registers,
tile partial sums exchanged through memory with the substep's sequence number in the fourth float.  Its contract: a call of n
substeps equals n tetsim_step calls (one tet and one particle kernel each) BIT FOR BIT, whatever n, and across calls."""
import numpy as np
import pytest

from conftest import load_mesh
from tetsim_amd import SoftBodyHIP, make_lattice

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _pair(v, t, **kw):
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    assert a.info.fused_particle_pass == 2, "expected the persistent frame kernel for a body of %d tets" % len(t)
    return a, b


@pytest.mark.parametrize("kw", [dict(), dict(constant_rest_shape=True)])
def test_frame_kernel_equals_stepwise_kernels_bit_for_bit(kw):
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])      # start just above the floor: contact and friction from the first frame
    a, b = _pair(v, t, **kw)
    total = 0
    for n in (20, 1, 2, 7, 20, 20, 33):          # odd and even counts: the partial-sum buffers alternate by substep parity
        a.simulateSubsteps(n, DT, PP)
        for _ in range(n):
            b.simulate(DT, PP)
        total += n
        assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and _same(a.quats, b.quats), "after %d substeps (last call: %d)" % (total, n)
    assert a.pos[:, 1].min() < 0.02             # the Dragon has landed: floor contact and friction were exercised


def test_frame_kernel_with_grab_dt_change_and_mixed_calls():
    v, t = make_lattice(12, y0=0.02)              # 10,368 tets = 41 tiles
    a, b = _pair(v, t)
    gid = 7
    for k, dt in enumerate((DT, DT * 2, DT, DT * 0.5)):
        if k == 1:
            for body in (a, b):
                body.setGrab(gid, [0.1, 0.9, -0.2])
        if k == 3:
            for body in (a, b):
                body.endGrab()
        a.simulateSubsteps(10, dt, PP)
        a.simulate(dt, PP)                        # a stepwise call between two frame calls: both paths share the state arrays
        a.simulateSubsteps(9, dt, PP)
        for _ in range(20):
            b.simulate(dt, PP)
        assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and _same(a.quats, b.quats), k
    assert np.isfinite(a.pos).all()


def test_frame_kernel_batch_and_save_load():
    """Several bodies behind one handle: tiles of different bodies never wait for each other; a checkpoint taken between two
    frame calls restores the trajectory bit for bit."""
    dv, dt_ = load_mesh("dragon")
    lv, lt = make_lattice(5, y0=0.1)
    bodies = [(dv, dt_), (lv, lt), (dv + np.float32([1.0, 0.2, 0.0]), dt_), (lv + np.float32([-1.0, 0.0, 0.5]), lt)]
    a = SoftBodyHIP.batch(bodies, dict(PP), solver="polar", precision="fast")
    b = SoftBodyHIP.batch(bodies, dict(PP), solver="polar", precision="fast")
    assert a.info.fused_particle_pass == 2
    a.simulateSubsteps(20, DT, PP)
    blob = a.saveState()
    a.simulateSubsteps(15, DT, PP)
    for _ in range(20):
        b.simulate(DT, PP)
    c = SoftBodyHIP.batch(bodies, dict(PP), solver="polar", precision="fast")
    c.loadState(blob)
    assert _same(c.pos, b.pos)
    c.simulateSubsteps(15, DT, PP)
    for _ in range(15):
        b.simulate(DT, PP)
    assert _same(a.pos, b.pos) and _same(c.pos, b.pos) and _same(a.quats, b.quats) and _same(c.quats, b.quats)


def test_large_bodies_keep_one_kernel_per_substep():
    v, t = make_lattice(40)                        # 384,000 tets = 1,500 tiles: more than half the device's resident workgroups
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    assert body.info.fused_particle_pass == 1
