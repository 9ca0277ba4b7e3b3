"""One launch per substep (pj_blocked.hip: pjb_substep_kernel): large unpartitioned FAST bodies run a substep as ONE
grid -- the tet kernel's tile workgroups, and behind them particle workgroups whose waves wait for the tiles their 64 particles
depend on and then run the particle kernel's arithmetic (same lists, same order of additions).  A body stepped this way must equal
the same body stepped with a tet kernel and a particle kernel per substep BIT FOR BIT: eager steps and replayed graphs, any number of
substeps per call, dt changes, floor contact, a grab, the constant-rest-shape formulation; a body with a particle no tet touches
keeps the two-kernel substep."""
import os

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


def _body(v, t, finish, **kw):
    old = {k: os.environ.get(k) for k in ("TETSIM_ONE_LAUNCH_SUBSTEP", "TETSIM_FUSED_PARTICLE_PASS")}
    os.environ["TETSIM_ONE_LAUNCH_SUBSTEP"] = "1" if finish else "0"
    os.environ["TETSIM_FUSED_PARTICLE_PASS"] = "0"      # the yardstick: a tet kernel and a particle kernel per substep
    try:
        b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    finally:
        for k, x in old.items():
            if x is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = x
    assert b.info.fused_particle_pass == (3 if finish else 0)
    return b


@pytest.mark.parametrize("mesh,kw", [("dragon", dict()), ("lat20", dict()), ("lat20", dict(constant_rest_shape=True))])
def test_tile_finished_substeps_equal_the_two_kernel_substeps_bit_for_bit(mesh, kw):
    if mesh == "dragon":
        v, t = load_mesh("dragon")
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
    else:
        v, t = make_lattice(20, y0=0.01)
    a, b = _body(v, t, True, **kw), _body(v, t, False, **kw)
    gid = len(v) // 3
    total = 0
    for k, (n, dt) in enumerate(((20, DT), (1, DT), (7, DT), (2, DT * 2), (20, DT), (3, DT * 0.5), (33, DT))):
        if k == 2:
            for x in (a, b):
                x.setGrab(gid, [float(v[gid, 0]) + 0.05, float(v[gid, 1]) + 0.1, float(v[gid, 2])])
        if k == 5:
            for x in (a, b):
                x.endGrab()
        a.simulateSubsteps(n, dt, PP)             # one graph replay per call
        if k % 2:
            b.simulateSubsteps(n, dt, PP)
        else:
            for _ in range(n):                    # ... against eager single steps
                b.simulate(dt, PP)
        total += n
        assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and _same(a.quats, b.quats), "after %d substeps (call %d)" % (total, k)
    assert np.isfinite(a.pos).all() and a.pos[:, 1].min() < 0.02


def test_eager_tile_finished_steps_equal_a_replayed_call():
    v, t = make_lattice(14, y0=0.01)
    a, b = _body(v, t, True), _body(v, t, True)
    a.simulateSubsteps(40, DT, PP)
    for _ in range(40):
        b.simulate(DT, PP)
    assert _same(a.pos, b.pos) and _same(a.quats, b.quats)


def test_headline_sized_body_uses_it_and_equals_the_two_kernel_substep():
    v, t = make_lattice(55)
    v = v - np.float32([0.0, v[:, 1].min() - 0.02, 0.0])     # (same tiling as the benchmark body, 2 cm above the floor)
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    assert a.info.fused_particle_pass == 3        # the default for bodies too large for the fused kernel
    b = _body(v, t, False)
    for n in (20, 20, 5):
        a.simulateSubsteps(n, DT, PP)
        b.simulateSubsteps(n, DT, PP)
    assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and _same(a.quats, b.quats)
    a.sync()                                      # (reports a partial sum that never became visible, if there was one)
    assert a.pos[:, 1].min() < 0.02


def test_a_body_with_a_loose_particle_keeps_the_two_kernel_substep():
    v, t = make_lattice(6, y0=0.3)
    v = np.concatenate([v, np.float32([[9.0, 9.0, 9.0]])])     # a particle no tet touches: no tile would ever finish its group
    old = os.environ.get("TETSIM_ONE_LAUNCH_SUBSTEP")
    os.environ["TETSIM_ONE_LAUNCH_SUBSTEP"] = "1"
    try:
        b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    finally:
        if old is None:
            os.environ.pop("TETSIM_ONE_LAUNCH_SUBSTEP", None)
        else:
            os.environ["TETSIM_ONE_LAUNCH_SUBSTEP"] = old
    assert b.info.fused_particle_pass != 3
    b.simulateSubsteps(5, DT, PP)
    assert np.isfinite(b.pos[:-1]).all()
