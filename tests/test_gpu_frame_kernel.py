"""The persistent frame kernel (pj_blocked.hip: pjb_frame_kernel): small unpartitioned FAST bodies run a whole
tetsim_step_n call as ONE launch -- every tile's workgroup resident for all n substeps, tet records and particles in registers,
tile partial sums exchanged through memory with the substep's sequence number in the fourth float.  Its contract: a call of n
substeps equals n tetsim_step calls (the same kernel for one substep) and equals the stepwise kernels (one tet and one particle kernel
per substep: what tetsim_profile steps with) BIT FOR BIT, whatever n, and across calls."""
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
    # 3: the four-lanes-per-tet frame kernel on 64-tet tiles (pj_quad.hip), the default for small bodies that carry their rest shape;
    # 2: the one-lane-per-tet one on 256-tet tiles (pj_blocked.hip) -- constant-rest-shape bodies
    assert a.info.fused_particle_pass == (2 if kw.get("constant_rest_shape") else 3), "expected the persistent frame kernel for a body of %d tets" % len(t)
    return a, b


@pytest.mark.parametrize("kw", [dict(), dict(constant_rest_shape=True)])
def test_frame_kernel_equals_stepwise_kernels_bit_for_bit(kw):
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])      # start just above the floor: contact and friction from the first frame
    a, b = _pair(v, t, **kw)
    c = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    total = 0
    for n in (20, 1, 2, 7, 20, 20, 33):          # odd and even counts: the partial-sum buffers alternate by substep parity
        a.simulateSubsteps(n, DT, PP)            # one persistent launch for the n substeps
        for _ in range(n):
            b.simulate(DT, PP)                   # one persistent launch per substep (tetsim_step)
        pr = c.profile(n, DT, PP)                # the stepwise kernels: a tet and a particle kernel per substep (256-tet tiles: tet | fused x (n - 1) | particle)
        assert pr["substeps"] == n and pr["tet_launches"] + pr["vertex_launches"] >= n
        total += n
        for other in (b, c):
            assert _same(a.pos, other.pos) and _same(a.vel, other.vel) and _same(a.quats, other.quats), "after %d substeps (last call: %d)" % (total, n)
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
    assert a.info.fused_particle_pass == 3
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


def test_many_independent_small_bodies_step_concurrently():
    """Every body has its own stream, and a persistent frame kernel needs all its workgroups resident at once: launches of DIFFERENT
    bodies that overlap on the device must not starve each other (the dispatcher hands out workgroups kernel by kernel; 40 bodies x 100
    substeps per call and 24 x 400 were run the same way on the builder's box: no wait ever gave up).  Twelve Dragons stepped round-robin
    without synchronising equal a Dragon stepped alone, bit for bit, and stay on the four-lane path."""
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
    solo = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    many = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast") for _ in range(12)]
    for _ in range(30):
        for b in many:
            b.simulateSubsteps(20, DT, PP)
    for _ in range(30):
        solo.simulateSubsteps(20, DT, PP)
    ref = solo.pos
    for b in many:
        assert _same(b.pos, ref) and b.info.fused_particle_pass == 3


def test_bodies_of_more_than_half_the_device_take_turns():
    """A body whose tiles need MORE than half the device's resident workgroups (131,712 tets = 515 tiles of 768) still runs a call as one
    persistent launch, equal to its stepwise kernels bit for bit; while it lives, the persistent launches of the device's bodies take
    turns: two such bodies and a Dragon stepped round-robin without synchronising equal the same bodies stepped alone."""
    v, t = make_lattice(28, y0=0.02)
    dv, dt_ = load_mesh("dragon")
    big = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast") for _ in range(3)]
    dragons = [SoftBodyHIP(dv, dt_, None, dict(PP), solver="polar", precision="fast") for _ in range(2)]
    assert all(b.info.fused_particle_pass == 2 for b in big) and all(d.info.fused_particle_pass == 3 for d in dragons)
    for n in (20, 3, 20, 20):
        big[0].simulateSubsteps(n, DT, PP)
        for _ in range(n):
            big[1].simulate(DT, PP)
    assert _same(big[0].pos, big[1].pos) and _same(big[0].quats, big[1].quats)
    for _ in range(10):                      # interleaved, no synchronisation in between
        big[1].simulateSubsteps(20, DT, PP)
        dragons[0].simulateSubsteps(20, DT, PP)
        big[2].simulateSubsteps(20, DT, PP)
    for _ in range(10):
        big[0].simulateSubsteps(20, DT, PP)
        dragons[1].simulateSubsteps(20, DT, PP)
    big[2].simulateSubsteps(63, DT, PP)      # (catches up with the 63 substeps the other two did first)
    assert _same(big[0].pos, big[1].pos) and _same(big[0].pos, big[2].pos) and _same(dragons[0].pos, dragons[1].pos)
    assert all(b.info.fused_particle_pass == 2 for b in big)


def test_an_exclusive_body_is_counted_from_its_creation():
    """Advisor, round 4: an exclusive body used to join its device's turn-taking at its own FIRST launch, so a persistent launch another
    body issued between its creation and that launch was untracked, and the two could sit half resident next to each other.  The
    body is counted when it is created: a Dragon's long call issued right after (5,000 substeps, milliseconds on the device) is
    already taking turns, the exclusive body's first call waits for it, and both equal bodies stepped alone.  When the exclusive body
    dies the count is given back (later launches no longer wait for each other: the twelve-Dragons test above is that state)."""
    v, t = make_lattice(28, y0=0.02)
    dv, dt_ = load_mesh("dragon")
    solo_big = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    solo_big.simulateSubsteps(20, DT, PP)
    ref_big = solo_big.pos
    solo_big.close()
    solo_dragon = SoftBodyHIP(dv, dt_, None, dict(PP), solver="polar", precision="fast")
    solo_dragon.simulateSubsteps(5000, DT, PP)
    ref_dragon = solo_dragon.pos
    solo_dragon.close()
    big = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")          # exclusive: counted here
    dragon = SoftBodyHIP(dv, dt_, None, dict(PP), solver="polar", precision="fast")
    assert big.info.fused_particle_pass == 2 and dragon.info.fused_particle_pass == 3
    dragon.simulateSubsteps(5000, DT, PP)      # in flight ...
    big.simulateSubsteps(20, DT, PP)           # ... when the exclusive body launches for the first time
    assert _same(big.pos, ref_big) and _same(dragon.pos, ref_dragon)
    big.close()
    dragon.simulateSubsteps(20, DT, PP)        # the count is back at zero: a plain launch
    assert np.isfinite(dragon.pos).all()


def test_an_exclusive_body_and_one_launch_calls_take_turns():
    """Round 6 (tools/soak.py found it): the one-launch calls of large bodies (pjb_call_kernel, nh_call_kernel: workgroups that WAIT for
    stamped data of workgroups dispatched before them) next to an exclusive persistent frame kernel (515 tiles that must ALL be resident)
    could each hold the slots the other needed -- 'a tile waited in vain' after the bounded wait, every call.  While an exclusive body
    lives, those launches take part in the device's turn-taking too: stepped round-robin without synchronising, every body equals its
    solo run bit for bit and no wait gives up (tetsim_sync would raise)."""
    lv, lt = make_lattice(28, y0=0.05)
    bv, bt = make_lattice(46, y0=0.05)

    makers = [lambda: SoftBodyHIP(lv, lt, None, dict(PP), solver="polar", precision="fast"),
              lambda: SoftBodyHIP(bv, bt, None, dict(PP), solver="polar", precision="fast"),
              lambda: SoftBodyHIP(bv, bt, None, dict(PP), solver="polar", precision="fast", lean_state=True),
              lambda: SoftBodyHIP(lv, lt, None, dict(PP), solver="neohookean", precision="fast", order="clustered")]
    calls = [int(n) for n in np.random.default_rng(1).integers(1, 41, size=40)]
    ref = []
    for make in makers:                      # each alone first (no other body alive: nothing to take turns with)
        body = make()
        for n in calls:
            body.simulateSubsteps(n, DT, PP)
        ref.append(body.pos)
        body.close()
    together = [make() for make in makers]
    assert [b.info.fused_particle_pass for b in together] == [2, 5, 5, 0]
    for n in calls:
        for b in together:
            b.simulateSubsteps(n, DT, PP)
    for b, r in zip(together, ref):
        b.sync()
        assert _same(b.pos, r)
    together[3].simulate(DT, PP)             # ... and tetsim_step of the one-launch sweep takes its turn too
    together[0].simulate(DT, PP)
    together[3].sync()
    together[0].sync()


def test_the_blocks_of_sequence_numbers_wrap():
    """Every call that hands data on by stamps takes a fresh block of 65,536 sequence numbers (tetsim_api.hip: next_epoch_block); after
    65,534 blocks the 32-bit count starts over.  70,000 calls of one substep take a body through the wrap; its twin makes the same 70,000
    substeps in 7,000 calls and stays far from it: the persistent frame kernel (Dragon, four lanes), the polar call kernel (486,680 tets)
    and the Gauss-Seidel call kernel end bit-equal to their twins, and no wait gives up."""
    dv, dt_ = load_mesh("dragon")
    bv, bt = make_lattice(46, y0=0.05)
    lv, lt = make_lattice(12, y0=0.05)
    for make, path in ((lambda: SoftBodyHIP(dv, dt_, None, dict(PP), solver="polar", precision="fast"), 3),
                       (lambda: SoftBodyHIP(bv, bt, None, dict(PP), solver="polar", precision="fast"), 5),
                       (lambda: SoftBodyHIP(lv, lt, None, dict(PP), solver="neohookean", precision="fast", order="clustered"), 0)):
        a, b = make(), make()
        assert a.info.fused_particle_pass == b.info.fused_particle_pass == path
        for _ in range(70000):
            a.simulateSubsteps(1, DT, PP)
        for _ in range(7000):
            b.simulateSubsteps(10, DT, PP)
        a.sync()
        b.sync()
        assert _same(a.pos, b.pos) and np.isfinite(a.pos).all(), path
        a.close()
        b.close()


def _wheel(spokes):
    """`spokes` tets around a common axis (particles 0 and 1): both axis particles have valence `spokes`."""
    ang = np.linspace(0.0, 2.0 * np.pi, spokes, endpoint=False)
    ring = np.stack([0.5 * np.cos(ang), np.full(spokes, 1.0), 0.5 * np.sin(ang)], axis=1)
    v = np.concatenate([[[0.0, 1.3, 0.0], [0.0, 0.7, 0.0]], ring]).astype(np.float32)
    i = np.arange(spokes)
    t = np.stack([np.zeros(spokes, int), np.ones(spokes, int), 2 + i, 2 + (i + 1) % spokes], axis=1).astype(np.int32)
    return v, t


def test_long_partial_lists_keep_the_256_tet_tiles():
    """The four-lane kernels take at most 12 partial sums per particle (host_prep.h kQuadMaxPartials).  A wheel of 1,000 tets around one
    axis puts its two axis particles into 16 tiles of 64 tets: such a body keeps the 256-tet tiles (4 tiles: lists of 4), says so, and
    its frame kernel still equals its stepwise kernels bit for bit; a wheel of 600 (10 tiles of 64) takes the four-lane path."""
    for spokes, want in ((1000, 2), (600, 3)):
        v, t = _wheel(spokes)
        a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_slot_table=False)
        b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_slot_table=False)
        assert a.info.fused_particle_pass == want, (spokes, a.info.fused_particle_pass)
        a.simulateSubsteps(25, DT, PP)
        for _ in range(25):
            b.simulate(DT, PP)
        assert _same(a.pos, b.pos) and _same(a.quats, b.quats) and np.isfinite(a.pos).all(), spokes


def test_lists_of_more_than_nine_partial_sums_stay_on_the_fused_paths():
    """The fused and frame kernels gather up to nine partial sums per particle in one trip and longer lists entry by entry behind them
    (irregular meshes; a 28^3-cell lattice has lists of ten).  A wheel of 3,600 tets puts its two axis particles into 15 tiles of 256
    tets: one persistent launch per call (mode 2), equal bit for bit to its stepwise fused kernel AND to the two-kernel substep of a
    process that has the fused pass switched off."""
    import os
    import subprocess
    import sys
    import tempfile
    v, t = _wheel(3600)
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_slot_table=False)
    b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_slot_table=False)
    assert a.info.fused_particle_pass == 2
    a.simulateSubsteps(25, DT, PP)
    for _ in range(25):
        b.simulate(DT, PP)
    assert _same(a.pos, b.pos) and _same(a.quats, b.quats) and np.isfinite(a.pos).all()
    with tempfile.TemporaryDirectory() as tmp:
        code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
                "from test_gpu_frame_kernel import _wheel, PP, DT; from tetsim_amd import SoftBodyHIP\n"
                "v, t = _wheel(3600); c = SoftBodyHIP(v, t, None, dict(PP), solver='polar', precision='fast', ref_slot_table=False)\n"
                "assert c.info.fused_particle_pass == 5\n"
                "c.simulateSubsteps(25, DT, PP); np.save(%r, c.pos)\n"
                % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), os.path.join(tmp, "two.npy")))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TETSIM_FUSED_PARTICLE_PASS="0", TETSIM_QUAD="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert _same(np.load(os.path.join(tmp, "two.npy")), a.pos)
    v, t = make_lattice(28)                        # 131,712 tets = 515 tiles, lists of up to ten: one fused kernel per substep or one launch per call
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    assert body.info.fused_particle_pass in (1, 2)


def test_quad_tiles_with_loose_particles_fall_back():
    """A particle no tet touches is summed by no tile: the fused / frame / four-lane paths cannot serve it (it integrates to NaN in the
    reference too, 0 / 0) and the body keeps the two-kernel substep -- also for a body small enough for the four-lane kernels."""
    v, t = make_lattice(3, y0=0.5)
    v2 = np.concatenate([v, [[2.0, 2.0, 2.0]]]).astype(np.float32)
    body = SoftBodyHIP(v2, t, None, dict(PP), solver="polar", precision="fast")
    assert body.info.fused_particle_pass == 5      # (two kernels per substep through tetsim_step; a call of tetsim_step_n is one launch of both)
    body.simulateSubsteps(5, DT, PP)
    assert np.isfinite(body.pos[:-1]).all()
