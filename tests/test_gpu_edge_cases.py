"""Edge cases through the C ABI: empty / ragged inputs, isolated particles, error behaviour (GPU)."""
import numpy as np
import pytest

from conftest import load_mesh
from oracle import OracleNH, OraclePJ
from tetsim_amd import SoftBodyHIP, TetSimError, make_lattice

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def test_tetless_body_neohookean_is_free_fall():
    """numElems == 0: invMass stays 0, particles fall and hit the floor (Softbody.js accepts it; volError = NaN)."""
    v, _ = load_mesh("notets")
    t = np.zeros((0, 4), dtype=np.int32)
    body, orc = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean"), OracleNH(v, t, PP)
    for _ in range(300):
        body.simulate(DT * 2, PP)
        orc.simulate(DT * 2, PP)
    assert np.array_equal(body.pos.view(np.uint32), orc.pos.view(np.uint32))
    assert np.isnan(body.volError) and np.isnan(orc.volError)
    assert body.pos[:, 1].min() == 0.0


@pytest.mark.parametrize("precision,kw", [("precise", {}), ("fast", {}), ("fast", dict(gather=True))])
def test_isolated_particles_polar_follow_the_restatement(precision, kw):
    """A particle touched by no tet averages zero goals: 0/0 = NaN in P5 (SoftbodyGPU.js:319); P6's clamp then maps the
    NaN onto the bounds (IEEE min/max drop the NaN operand; GLSL leaves it undefined).  All device formulations must
    treat those particles exactly like the CPU restatement and leave the others unharmed."""
    v, t = make_lattice(3, y0=0.2)
    v = np.concatenate([v, [[0.1, 3.0, 0.1], [0.2, 3.0, 0.2]]]).astype(np.float32)  # two particles no tet references
    body, orc = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, **kw), OraclePJ(v, t, PP)
    for _ in range(10):
        body.simulate(DT, PP)
        orc.simulate(DT, PP)
    p, q = body.pos, orc.pos
    assert np.isfinite(p).all() and np.isfinite(q).all()
    assert np.abs(p[-2:] - q[-2:]).max() < 1e-6 and np.all(p[-2:, 1] == 0.0)   # clamped to the box, resting on the floor
    assert np.abs(p[:-2] - q[:-2]).max() < 1e-5


def test_polar_tetless_body_creates_and_steps():
    v, _ = load_mesh("notets")
    t = np.zeros((0, 4), dtype=np.int32)
    body, orc = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast"), OraclePJ(v, t, PP)
    for _ in range(3):
        body.simulate(DT, PP)
        orc.simulate(DT, PP)
    assert np.abs(body.pos - orc.pos).max() < 1e-6  # every particle is isolated: all sit clamped in the box corner


def test_bad_arguments():
    v, t = make_lattice(2)
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar")
    with pytest.raises(TetSimError):
        body.simulate(0.0, PP)           # dt must be positive
    with pytest.raises(TetSimError):
        body.simulate(float("nan"), PP)
    with pytest.raises(TetSimError):
        body.setGrab(10 ** 6, [0, 0, 0])  # particle id out of range
    with pytest.raises(TetSimError):
        _ = body.volError                # Neo-Hookean only
    nh = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean")
    with pytest.raises(TetSimError):
        _ = nh.quats                     # polar only
    with pytest.raises(TetSimError):
        SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", part_count=2, part_index=0)
    with pytest.raises(TetSimError):
        SoftBodyHIP(v, t, None, dict(PP), solver="polar", device=99)


def test_write_state_round_trip_and_repredict():
    """Checkpoint/restore: state read from one body and written into a fresh one continues identically (PRECISE)."""
    v, t = load_mesh("dragon")
    a = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean")
    for _ in range(15):
        a.simulate(DT, PP)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean")
    b.writeState(a.pos, a.vel)
    for _ in range(15):
        a.simulate(DT, PP)
        b.simulate(DT, PP)
    assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32))


def test_many_bodies_share_the_device():
    """Several handles (each with its own stream) advance independently, like softBodies[] in main.js:80-84."""
    v, t = load_mesh("dragon")
    bodies = [SoftBodyHIP(v + np.float32(0.01 * i), t, None, dict(PP), solver="polar", precision="fast") for i in range(4)]
    for _ in range(3):
        for b in bodies:
            b.simulateSubsteps(20, DT, PP)
    ref = SoftBodyHIP(v + np.float32(0.02), t, None, dict(PP), solver="polar", precision="fast")
    for _ in range(3):
        ref.simulateSubsteps(20, DT, PP)
    assert np.array_equal(bodies[2].pos.view(np.uint32), ref.pos.view(np.uint32))


def test_device_grab_query_matches_reference_argmin():
    """tetsim_start_grab runs the argmin of Softbody.js:279-291 on the device (f64 distances, first minimum wins),
    including exact ties (symmetric lattice points) and after the internal Morton renumbering of the polar path."""
    v, t = make_lattice(6, y0=0.1)
    orc = OracleNH(v, t, PP)
    bodies = [SoftBodyHIP(v, t, None, dict(PP), solver="neohookean"), SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")]
    h = 1.0 / 6
    queries = [(0.0, 0.6, 0.0), (0.5 * h, 0.1 + 0.5 * h, 0.5 * h), (-0.5, 0.1, -0.5), (3.0, 3.0, 3.0), (0.25 * h, 0.1 + 2.5 * h, -1.5 * h),
               (0.013, 0.47, -0.21)]
    for q in queries:
        want = orc.startGrab(*[float(np.float32(c)) for c in q])   # the ABI carries the query point as f32
        for b in bodies:
            assert b.startGrab(np.float32(q)) == want, q
    # after motion the query must see the LATEST positions
    for b in bodies:
        b.endGrab(); b.simulateSubsteps(30, DT, PP)
    b = bodies[0]
    orc2 = OracleNH(b.pos, t, PP)
    assert b.startGrab(np.float32([0.1, 0.3, 0.1])) == orc2.startGrab(float(np.float32(0.1)), float(np.float32(0.3)), float(np.float32(0.1)))


def test_nearest_particle_over_partitions_equals_start_grab():
    """tetsim_nearest_particle on each partition + a host-side min reproduces startGrab of the monolithic body (the device
    argmin of Softbody.js:279-291), also after the body has moved; then the grab pins the particle on its owner only."""
    from conftest import load_mesh
    from tetsim_amd import group_step_n
    v, t = load_mesh("dragon")
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0)
    owner = (np.arange(len(v)) * 3 // len(v)).astype(np.int32)
    for solver_kw in (dict(solver="polar", precision="fast"), dict(solver="polar", precision="precise")):
        mono = SoftBodyHIP(v, t, None, dict(pp), **solver_kw)
        parts = [SoftBodyHIP(v, t, None, dict(pp), part_count=3, part_index=i, vert_owner=owner, **solver_kw) for i in range(3)]
        mono.simulateSubsteps(15, 1 / 1200, pp)
        group_step_n(parts, 15, 1 / 1200, pp)
        rng = np.random.RandomState(3)
        for q in np.vstack([rng.uniform(-1, 2, (20, 3)), v[[0, 500, 1233]] + 1e-4]):
            want = mono.startGrab(q)
            cands = [p.nearestParticle(q) for p in parts]
            gid = min(cands, key=lambda c: (c[1], c[0]))[0]
            assert gid == want, (q, cands, want)
        target = np.float32([0.3, 1.5, 0.2])
        for p in parts:
            p.setGrab(gid, target)
        group_step_n(parts, 1, 1 / 1200, pp)
        for p in parts:
            ids = p.ownedIds
            hit = np.nonzero(ids == gid)[0]
            if len(hit):
                assert np.allclose(p.pos[hit[0]], target, atol=1e-6)


@pytest.mark.parametrize("solver,precision", [("polar", "fast"), ("polar", "precise"), ("neohookean", "precise")])
def test_pinned_zero_copy_readback_equals_copying_readback(solver, precision):
    v, t = load_mesh("dragon")
    b = SoftBodyHIP(v, t, None, dict(PP), solver=solver, precision=precision)
    b.simulateSubsteps(10, DT, PP)
    view = b.posPinned
    assert np.array_equal(view.view(np.uint32), b.pos.view(np.uint32))
    addr = view.ctypes.data
    b.simulateSubsteps(10, DT, PP)
    view2 = b.posPinned
    assert view2.ctypes.data == addr                      # same pinned memory, refreshed in place
    assert np.array_equal(view2.view(np.uint32), b.pos.view(np.uint32))
