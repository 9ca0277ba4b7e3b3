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


@pytest.mark.parametrize("precision", ["precise", "fast"])
def test_batch_runs_one_workgroup_per_body(precision):
    """tetsim_create_batch: every body of the batch gets a workgroup (and a CU's LDS) of its own in the one launch of a call.  Each body
    equals its solo run bit for bit through tetsim_step_n and tetsim_step, and the stepwise kernels (tetsim_profile) agree with both."""
    meshes = []
    for name, lift in (("dragon", 0.0), ("lat4", 0.0), ("dragon", -0.3), ("lat4", 0.5)):
        v, t = load_mesh(name)
        meshes.append((v + np.float32([0.0, lift, 0.0]), t))
    lv, lt = make_lattice(6, y0=0.0002)          # on the floor within the first substeps
    meshes.append((lv, lt))
    kw = dict(solver="neohookean", order="coloured", precision=precision)
    batch = SoftBodyHIP.batch(meshes, dict(PP), **kw)
    twin = SoftBodyHIP.batch(meshes, dict(PP), **kw)
    solos = [SoftBodyHIP(v, t, None, dict(PP), **kw) for v, t in meshes]
    assert batch.info.fused_particle_pass == 4 and batch.info.num_bodies == len(meshes)
    for n in (10, 3):
        for body in [batch] + solos:
            body.simulateSubsteps(n, DT, PP)
            body.simulate(DT, PP)
        twin.profile(n + 1, DT, PP)
    pos, vel = batch.pos, batch.vel
    assert _same(pos, twin.pos) and _same(vel, twin.vel) and batch.volError == twin.volError
    for (prange, _), solo in zip(batch.bodyRanges, solos):
        assert _same(pos[prange[0]:prange[1]], solo.pos) and _same(vel[prange[0]:prange[1]], solo.vel)
    assert np.isfinite(pos).all() and pos[:, 1].min() == 0.0


def test_batch_of_dragons_throughput():
    """The Dragon through the reference's CPU algorithm is one workgroup's chain of levels; 64 of them in one batch are 64 workgroups on 64
    CUs: >= 20x the tet-solves/s of a single one, bit-exact arithmetic."""
    import time
    v, t = load_mesh("dragon")

    def rate(body, frames):
        body.simulateSubsteps(10, DT, PP); body.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            body.simulateSubsteps(10, DT, PP)
        body.sync()
        return body.info.num_elems * 10 * frames / (time.perf_counter() - t0)

    kw = dict(solver="neohookean", order="coloured", precision="precise")
    one = rate(SoftBodyHIP(v, t, None, dict(PP), **kw), 50)
    many = rate(SoftBodyHIP.batch([(v, t)] * 64, dict(PP), **kw), 50)
    print("single Dragon %.1f M tet-solves/s, 64 Dragons in one batch %.1f M (%.1fx)" % (one / 1e6, many / 1e6, many / one))
    assert many >= 20.0 * one, (one, many)
