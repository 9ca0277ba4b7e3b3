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


@pytest.mark.parametrize("kw", [dict(precision="precise"), dict(precision="fast"), dict(precision="fast", gather=True),
                                dict(precision="fast", constant_rest_shape=True), dict(precision="fast", lean_state=True)])
def test_polar_save_load_state_continues_bit_for_bit(kw):
    """tetsim_save_state / tetsim_load_state carry the COMPLETE polar state (positions, velocities, quaternions, carried rest
    shape): a fresh body restored from the blob continues the original trajectory bit for bit -- which tetsim_write_state
    (positions + velocities only) cannot do, and says so."""
    v, t = load_mesh("dragon")
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", **kw)
    a.simulateSubsteps(30, DT, PP)
    blob = a.saveState()
    pos30, vel30 = a.pos, a.vel
    a.simulateSubsteps(25, DT, PP)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", **kw)
    b.loadState(blob)
    assert np.array_equal(b.pos.view(np.uint32), pos30.view(np.uint32))
    b.simulateSubsteps(25, DT, PP)
    assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32))
    assert np.array_equal(a.quats.view(np.uint32), b.quats.view(np.uint32))
    # positions + velocities alone are NOT the state of this solver: the quaternions / carried shape of a fresh body differ
    c = SoftBodyHIP(v, t, None, dict(PP), solver="polar", **kw)
    c.writeState(pos30, vel30)
    c.simulateSubsteps(25, DT, PP)
    assert not np.array_equal(a.quats.view(np.uint32), c.quats.view(np.uint32))
    # blobs are validated: other options, truncation, garbage
    other = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise", ref_slot_table=False)
    for bad in (blob[:100], b"\0" * len(blob), blob[:-16]):
        with pytest.raises(TetSimError):
            b.loadState(bad)
    if kw != dict(precision="precise"):
        with pytest.raises(TetSimError, match="another mesh or other options"):
            other.loadState(blob)


def test_state_blob_of_another_mesh_with_the_same_counts_is_rejected():
    """The header's counts and options cannot tell two meshes of the same size apart; its mesh digest can (a blob of the mirrored
    Dragon would otherwise overwrite quaternions and carried shapes while volumes and weights stayed this body's)."""
    v, t = load_mesh("dragon")
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    a.simulateSubsteps(5, DT, PP)
    blob = a.saveState()
    v2 = v.copy()
    v2[:, 0] *= 1.25                      # same particle and tet counts, same options, another shape
    b = SoftBodyHIP(v2, t, None, dict(PP), solver="polar", precision="fast")
    with pytest.raises(TetSimError, match="another mesh"):
        b.loadState(blob)
    c = SoftBodyHIP(v, t, None, dict(PP, density=500.0), solver="neohookean", order="coloured")
    d = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured")
    with pytest.raises(TetSimError, match="another mesh"):
        d.loadState(c.saveState())        # the density is part of what the blob belongs to (inverse masses)


@pytest.mark.parametrize("order,cells", [("clustered", 3), ("coloured", 3), ("coloured", 17)])
def test_collapsed_tet_does_not_poison_the_fast_sweeps(order, cells):
    """A tet whose four corners coincide has tr(F^T F) = 0: Softbody.js leaves it alone (C == 0 returns early, :176); the FAST
    kernels' rsq(0) * 0 must not turn that into NaN positions -- the four-lane cluster kernel, the single-workgroup launch of small
    bodies (one lane and four lanes per tet) and the level kernels of larger ones (17^3 cells: 5,832 particles)."""
    v, t = make_lattice(cells, y0=0.3)
    # collapse ALL of one tet's corners onto one point, in the current positions only (the rest pose stays regular)
    body = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order=order)
    pos = body.pos
    for k in (0, len(t) - 1):      # the first of the sweep (a wide level) and the last (a narrow one)
        pos[t[body.tetOrder[k]]] = pos[t[body.tetOrder[k], 0]]
    body.writeState(pos, np.zeros_like(pos))
    body.simulateSubsteps(3, DT, PP)
    assert np.isfinite(body.pos).all()
    body.simulate(DT, PP)
    assert np.isfinite(body.pos).all()


def test_neohookean_save_load_state():
    v, t = load_mesh("dragon")
    a = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured")
    a.simulateSubsteps(12, DT * 2, PP)
    blob = a.saveState()
    a.simulateSubsteps(9, DT * 2, PP)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured")
    b.loadState(blob)
    b.simulateSubsteps(9, DT * 2, PP)
    assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32)) and a.volError == b.volError


@pytest.mark.parametrize("kw", [dict(precision="precise"), dict(precision="fast"), dict(precision="fast", lean_state=True)])
def test_pinned_quaternion_view_equals_copying_read(kw):
    """SURVEY.md 8(f)-2, quaternion half: tetsim_read_quats_pinned is a view of a pinned host buffer (same address every
    call) and holds what tetsim_read_quats returns."""
    v, t = load_mesh("dragon")
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", **kw)
    body.simulateSubsteps(20, DT, PP)
    q1 = body.quatsPinned
    addr = q1.ctypes.data
    assert np.array_equal(q1.view(np.uint32), body.quats.view(np.uint32)) and q1.shape == (len(t), 4)
    body.simulateSubsteps(5, DT, PP)
    q2 = body.quatsPinned
    assert q2.ctypes.data == addr and np.array_equal(q2.view(np.uint32), body.quats.view(np.uint32))
    with pytest.raises(TetSimError):
        SoftBodyHIP(v, t, None, dict(PP), solver="neohookean").quatsPinned


def test_library_info_on_the_gpu_box():
    from tetsim_amd import library_info
    info = library_info()
    assert info["abi"] == 5 and info["ablation"] is False and len(info["source_sha"]) == 16


def test_partitioned_bodies_keep_their_rows_of_the_visual_mesh(tmp_path):
    """A partition owns a subset of the particles; of a visual mesh handed to it (arrays or a .tetsim file that stores one) it keeps
    the rows whose tet it owns -- both partitions together: every row once (tests/test_gpu_partition_state.py skins them)."""
    from conftest import load_f32
    from tetsim_amd.meshfile import write_mesh
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32")
    part = SoftBodyHIP(v, t, None, dict(PP), vis, solver="polar", precision="fast", part_count=2, part_index=1)
    assert 0 < part.info.num_vis_verts < len(vis) // 4 and part.info.owned_particles < len(v)
    path = str(tmp_path / "d.tetsim")
    write_mesh(path, v, t, vis_verts=vis)
    part2 = SoftBodyHIP.fromFile(path, dict(PP), solver="polar", precision="fast", part_count=2, part_index=0)
    part2.simulate(DT, PP)
    assert part2.info.num_vis_verts + part.info.num_vis_verts == len(vis) // 4 and np.isfinite(part2.pos).all()
    assert len(np.intersect1d(part.visualIds, part2.visualIds)) == 0
    whole = SoftBodyHIP.fromFile(path, dict(PP), solver="polar", precision="fast")
    assert whole.info.num_vis_verts == len(vis) // 4 and np.array_equal(whole.visualIds, np.arange(len(vis) // 4))


@pytest.mark.parametrize("kw", [dict(solver="polar", precision="precise"), dict(solver="polar", precision="fast"),
                                dict(solver="polar", precision="fast", gather=True), dict(solver="polar", precision="fast", constant_rest_shape=True),
                                dict(solver="polar", precision="fast", lean_state=True),
                                dict(solver="neohookean", precision="precise", order="original"), dict(solver="neohookean", precision="precise", order="coloured"),
                                dict(solver="neohookean", precision="precise", order="clustered"), dict(solver="neohookean", precision="fast", order="clustered")])
def test_batch_of_bodies_equals_solo_runs_bit_for_bit(kw):
    """tetsim_create_batch: several independent bodies behind one handle, one launch per kernel for all of them.  Each body's
    positions and velocities must equal its solo run bit for bit -- different meshes in one batch, floor contact included,
    through tetsim_step and tetsim_step_n."""
    meshes = []
    for name, y_min in (("dragon", None), ("lat4", None), ("hub", 0.25), ("dragon", 0.004), ("lat4", 0.8)):
        v, t = load_mesh(name)
        v = v.copy()
        if y_min is not None:
            v[:, 1] += np.float32(y_min) - v[:, 1].min()   # the second dragon starts 4 mm above the floor: contact within the run
        meshes.append((v, t))
    dt = DT if kw["solver"] == "polar" else DT * 2
    batch = SoftBodyHIP.batch(meshes, dict(PP), **kw)
    assert batch.info.num_bodies == len(meshes) and batch.info.num_particles == sum(len(v) for v, _ in meshes)
    solos = [SoftBodyHIP(v, t, None, dict(PP), **kw) for v, t in meshes]
    for body in [batch] + solos:
        body.simulateSubsteps(40, dt, PP)
        for _ in range(5):
            body.simulate(dt, PP)
        body.simulateSubsteps(40, dt, PP)
    pos, vel = batch.pos, batch.vel
    ranges = batch.bodyRanges
    assert [r[0][1] - r[0][0] for r in ranges] == [len(v) for v, _ in meshes] and [r[1][1] - r[1][0] for r in ranges] == [len(t) for _, t in meshes]
    for (prange, _), solo in zip(ranges, solos):
        assert np.array_equal(pos[prange[0]:prange[1]].view(np.uint32), solo.pos.view(np.uint32))
        assert np.array_equal(vel[prange[0]:prange[1]].view(np.uint32), solo.vel.view(np.uint32))
    assert pos[:, 1].min() == 0.0          # somebody is on the floor
    if kw["solver"] == "polar":
        assert batch.info.dropped_slots == sum(s.info.dropped_slots for s in solos)   # the slot-table quirk, once per body
    with pytest.raises(TetSimError, match="partitioned"):
        SoftBodyHIP.batch(meshes, dict(PP), part_count=2, part_index=0, **kw)


def test_batch_of_dragons_throughput():
    """Config 2's body is launch-bound alone (15 tiles on a chip with 2,048 workgroup slots): 64 of them in one batch must step at
    >= 20x the tet-solves/s of a single one."""
    import time
    v, t = load_mesh("dragon")

    def rate(body, frames):
        body.simulateSubsteps(20, DT, PP); body.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            body.simulateSubsteps(20, DT, PP)
        body.sync()
        return body.info.num_elems * 20 * frames / (time.perf_counter() - t0)

    one = rate(SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast"), 200)
    many = rate(SoftBodyHIP.batch([(v, t)] * 64, dict(PP), solver="polar", precision="fast"), 200)
    print("single Dragon %.1f M tet-solves/s, 64 Dragons in one batch %.1f M (%.1fx)" % (one / 1e6, many / 1e6, many / one))
    assert many >= 20.0 * one, (one, many)
