"""TETSIM_FLAG_LEAN_STATE (include/tetsim.h; pj_blocked.hip: kModeLeanState): the polar tet record without what FAST arithmetic does not
need streamed -- three corners of the carried shape instead of four (it is relative to its own centroid) and no quaternion (pure
output of a substep; recovered from the carried shape when it is read).  92 instead of 148 B per tet.

Contract: inside the same FAST envelope against the oracle (which carries four corners and multiplies the quaternion up, as the
reference does: SoftbodyGPU.js:181, :253-262) as the default FAST path; every bit-equality the default FAST path offers (a call of n
substeps == n calls of one == the stepwise kernels, batch == solo, save / load: tests/test_gpu_edge_cases.py,
test_gpu_partition_state.py); the read-out quaternions equal the multiplied-up ones to rounding, sign included.
The six GLSL goldens, the 1 M-tet lattice, the eight-slab checkpoint and the skinned normals run this mode in their own files
(parametrised there).
"""
import numpy as np
import pytest

from conftest import load_mesh, within
from oracle import OraclePJ
from tetsim_amd import SoftBodyHIP, TetSimError, group_step_n, make_lattice

pytestmark = pytest.mark.gpu

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT20 = (1.0 * (1.0 / 60.0)) / 20


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def _lean(v, t, **kw):
    return SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", lean_state=True, **kw)


def _input_order(body):
    q = body.quats
    out = np.empty_like(q)
    out[body.localTets] = q
    return out


def test_lean_state_tracks_the_oracle_and_the_default_fast_path():
    v, t = load_mesh("dragon")
    body, ref = _lean(v, t), SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    orc = OraclePJ(v, t, PP, slot_quirk=True)
    tol = {1: 2e-6, 20: 5e-5, 200: 2e-3}
    for step in range(1, 201):
        body.simulate(DT20, PP)
        ref.simulate(DT20, PP)
        orc.simulate(DT20, PP)
        if step in tol:
            within("polar fast lean vs oracle dragon @%d" % step, np.abs(body.pos - orc.pos).max(), tol[step])
            within("polar fast lean vs carried dragon @%d" % step, np.abs(body.pos - ref.pos).max(), tol[step])
            q = _input_order(body)
            assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-6
            # the recovered quaternion against the one the oracle multiplied up substep by substep: same rotation, SAME SIGN
            within("polar fast lean vs oracle dragon @%d (quat)" % step, np.abs(q - orc.quats).max(), max(50 * tol[step], 1e-5))
    assert body.info.fused_particle_pass == 2      # the 256-tet-tile persistent kernel (the four-lane kernels keep the reference's record)


def test_lean_state_call_of_n_equals_n_calls_equals_stepwise_kernels():
    """Small body: the persistent frame kernel (tetsim_step_n), the same kernel per substep (tetsim_step) and the stepwise tet / fused /
    particle kernels (tetsim_profile) share one arithmetic -- positions, velocities and the recovered quaternions bit for bit."""
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
    a, b, c = _lean(v, t), _lean(v, t), _lean(v, t)
    total = 0
    for n in (20, 1, 2, 7, 20, 33):
        a.simulateSubsteps(n, DT20, PP)
        for _ in range(n):
            b.simulate(DT20, PP)
        c.profile(n, DT20, PP)
        total += n
        for other in (b, c):
            assert _same(a.pos, other.pos) and _same(a.vel, other.vel), "after %d substeps (last call: %d)" % (total, n)
        assert _same(a.quats, b.quats) and _same(a.quats, c.quats)
    assert a.pos[:, 1].min() < 0.02


@pytest.mark.parametrize("cells,path", [(20, 2), (40, 1), (46, 5)])
def test_lean_state_mid_sized_and_large_bodies(cells, path):
    """48,000 tets: one persistent launch per call over all XCDs; 384,000 tets: one fused kernel per substep; 584,016 tets: tet + particle
    kernel per substep through tetsim_step, the whole call as one launch through tetsim_step_n (the headline's path).  step_n (graph) equals step by step, and the body stays inside the default FAST path's
    envelope against the oracle."""
    v, t = make_lattice(cells, y0=0.3)
    a, b = _lean(v, t), _lean(v, t)
    assert a.info.fused_particle_pass == path
    a.simulateSubsteps(20, DT20, PP)
    for _ in range(20):
        b.simulate(DT20, PP)
    assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and _same(a.quats, b.quats)
    orc = OraclePJ(v, t, PP, slot_quirk=True)
    for _ in range(20):
        orc.simulate(DT20, PP)
    within("polar fast lean vs oracle lattice %d^3 @20" % cells, np.abs(a.pos - orc.pos).max(), 2e-4)
    within("polar fast lean vs oracle lattice %d^3 @20 (quat)" % cells, np.abs(_input_order(a) - orc.quats).max(), 1e-3)


def test_lean_state_quaternion_keeps_its_sign_through_a_full_turn():
    """A free body spinning at 3 rad/s, read out every 20 substeps: the multiplied-up quaternion passes w = 0 after half a turn and
    carries on to -1; the recovered one is given the sign next to the previous read-out and does the same.  (Read out only once per
    more-than-half-a-turn it could not know -- include/tetsim.h says so.)"""
    v, t = make_lattice(6, y0=1.0)
    pp = dict(PP, gravity=0.0, worldBounds=[-50.0, -50.0, -50.0, 50.0, 50.0, 50.0])
    c = v.mean(axis=0)
    w = np.array([0.4, 0.3, 3.0])
    vel = np.cross(w, v - c).astype(np.float32)
    e0 = np.linalg.norm(v[t[:, 0]] - v[t[:, 1]], axis=1).astype(np.float64)
    lean = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", lean_state=True, ref_fixed_bounds=False, ref_slot_table=False)
    ref = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", ref_fixed_bounds=False, ref_slot_table=False)
    lean.writeState(v, vel)
    ref.writeState(v, vel)
    worst, wmin = 0.0, 1.0
    for _ in range(150):              # 3,000 substeps = 2.5 s: ~0.9 turns (one Jacobi iteration per substep bleeds some spin: 2.4 rad/s effective)
        lean.simulateSubsteps(20, DT20, pp)
        ref.simulateSubsteps(20, DT20, pp)
        ql, qr = _input_order(lean), _input_order(ref)
        worst = max(worst, float(np.abs(ql - qr).max()))
        wmin = min(wmin, float(qr[:, 3].min()))
    assert wmin < -0.2                 # the body has turned past pi: q.w went negative in the multiplied-up quaternion
    # two FAST trajectories of a free spinning body drift apart in PHASE over 3,000 substeps (observed 0.04 = 0.07 rad of 5.9); a lost
    # sign would read 2.  The recovery's own accuracy is what the oracle comparisons above and in the GLSL goldens bound.
    within("polar fast lean vs carried spinning lattice: quaternion over 3000 substeps", worst, 0.2)
    p = lean.pos
    e1 = np.linalg.norm(p[t[:, 0]] - p[t[:, 1]], axis=1)
    within("polar fast lean spinning lattice: relative edge drift after 3000 substeps", float(np.abs(e1 / e0 - 1.0).max()), 1e-3)


def test_lean_state_partitions_and_transports():
    """Eight slabs in one process (copy transport) against the monolithic lean body: ghost tets evolve their three corners identically
    on both sides of a cut; tolerance-level only through the tiles' summation order, like the default FAST path."""
    cells, parts = 16, 4
    v, t = make_lattice(cells, y0=0.02)
    owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * parts // (cells + 1), parts - 1).astype(np.int32)
    group = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", lean_state=True, part_count=parts, part_index=p, vert_owner=owner)
             for p in range(parts)]
    mono = _lean(v, t)
    group_step_n(group, 40, DT20, PP)
    mono.simulateSubsteps(40, DT20, PP)
    pos = np.empty_like(v)
    for b in group:
        pos[b.ownedIds] = b.pos
    within("polar fast lean 4 slabs vs monolithic lat16 @40", np.abs(pos - mono.pos).max(), 2e-4)
    for b in group:
        assert np.abs(np.linalg.norm(b.quats, axis=1) - 1.0).max() < 1e-5


def test_lean_state_zero_volume_tet_and_flag_checks():
    v, t = load_mesh("lat4")
    v = v.copy()
    t = t.copy()
    v[t[5, 3]] = (v[t[5, 0]] + v[t[5, 1]] + v[t[5, 2]]) / np.float32(3.0)      # tet 5 collapses into the plane of its first three corners
    body = _lean(v, t, ref_slot_table=False)
    dflt = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", ref_slot_table=False)
    body.simulateSubsteps(40, DT20, PP)
    dflt.simulateSubsteps(40, DT20, PP)
    q = _input_order(body)
    assert np.isfinite(dflt.pos).all() and np.isfinite(body.pos).all() and np.isfinite(q).all()   # (a zero-volume tet weighs nothing: SoftbodyGPU.js:220)
    assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-5
    within("polar fast lean vs carried lat4 with a collapsed tet @40", np.abs(body.pos - dflt.pos).max(), 1e-4)
    v, t = load_mesh("dragon")
    for kw in (dict(precision="precise"), dict(precision="fast", gather=True), dict(solver="neohookean"), dict(precision="fast", constant_rest_shape=True),
               dict(precision="fast", deep_ghosts=True)):
        args = dict(solver="polar")
        args.update(kw)
        with pytest.raises(TetSimError, match="LEAN_STATE"):
            SoftBodyHIP(v, t, None, dict(PP), lean_state=True, **args)


@pytest.mark.parametrize("kw", [dict(), dict(lean_state=True), dict(constant_rest_shape=True)])
def test_one_launch_substep_equals_two_kernels_bit_for_bit(kw):
    """Large unpartitioned bodies (>= 2,048 tiles): inside tetsim_step_n a substep is ONE launch -- every tile, then every particle, whose
    waves look for the substep's sequence number in their tiles' partial sums (pj_blocked.hip: pjb_substep_kernel).  Same arithmetic as
    the tet kernel + particle kernel pair tetsim_step launches: bit for bit, across calls of odd lengths, floor contact, a grab, a dt
    change; TETSIM_PJ_ONE_LAUNCH=0 (read at creation) keeps the pair inside the graphs too."""
    import os
    v, t = make_lattice(46, y0=0.01)             # 584,016 tets = 2,282 tiles
    a = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    os.environ["TETSIM_PJ_ONE_LAUNCH"] = "0"
    try:
        b = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    finally:
        del os.environ["TETSIM_PJ_ONE_LAUNCH"]
    c = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", **kw)
    assert a.info.fused_particle_pass == 5 and b.info.fused_particle_pass == 0
    for k, (n, dt) in enumerate(((20, DT20), (1, DT20), (7, DT20), (3, DT20 * 2), (20, DT20))):
        if k == 2:
            for body in (a, b, c):
                body.setGrab(11, [0.1, 0.5, -0.1])
        if k == 4:
            for body in (a, b, c):
                body.endGrab()
        a.simulateSubsteps(n, dt, PP)            # graphs of one-launch substeps
        b.simulateSubsteps(n, dt, PP)            # graphs of kernel pairs
        for _ in range(n):
            c.simulate(dt, PP)                   # tetsim_step: eager kernel pairs
        assert _same(a.pos, b.pos) and _same(a.pos, c.pos) and _same(a.vel, c.vel), (kw, k)
    assert _same(a.quats, c.quats) and a.pos[:, 1].min() == 0.0


def test_two_one_launch_calls_side_by_side():
    """Two large bodies on their own streams, both stepping whole calls as one launch of stamped hand-overs, without synchronising in
    between: each kernel's waiting workgroups only ever wait for workgroups of the SAME kernel dispatched before them, so the two cannot
    starve each other -- each equals its solo run bit for bit and no wait gives up (tetsim_sync would say so)."""
    v, t = make_lattice(46, y0=0.02)
    solo = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    pair = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast"), SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", lean_state=True),
            SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")]
    for _ in range(12):
        for b in pair:
            b.simulateSubsteps(20, DT20, PP)
    for _ in range(12):
        solo.simulateSubsteps(20, DT20, PP)
    for b in pair:
        b.sync()
    assert _same(pair[0].pos, solo.pos) and _same(pair[2].pos, solo.pos) and np.isfinite(pair[1].pos).all()
    assert all(b.info.fused_particle_pass == 5 for b in pair)
