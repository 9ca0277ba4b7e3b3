"""The clustered Gauss-Seidel sweep as ONE launch per substep (nh_kernels.inc: nh_sweep1_kernel; dev_common.h: NHSweep): the clusters of all
colours in one grid, colour by colour, every particle handed from a cluster to the next one that touches it as one 16-byte store
{x, y, z, stamp}.  The arithmetic is the one-launch-per-colour sweep's, operation for operation, and the order of the solves is the
sequential order of Softbody.js:207-208 fed tetIds[tetsim_get_tet_order()] either way -- so the two must agree BIT FOR BIT: through
tetsim_step, tetsim_step_n (graph replay), mixed call lengths, floor contact, a grab, dt changes, and after a checkpoint.
(TETSIM_NH_ONE_LAUNCH=0, read when a body is created, keeps one launch per colour: the twin of every test here.)"""
import os

import numpy as np
import pytest

from conftest import load_mesh, within
from oracle import OracleNH
from tetsim_amd import SoftBodyHIP, make_lattice

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 10


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def _pair(v, t, precision="fast", **kw):
    one = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision=precision, order="clustered", **kw)
    os.environ["TETSIM_NH_ONE_LAUNCH"] = "0"
    try:
        per_colour = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision=precision, order="clustered", **kw)
    finally:
        del os.environ["TETSIM_NH_ONE_LAUNCH"]
    return one, per_colour


@pytest.mark.parametrize("mesh", ["dragon", "lat12", "lattice30", "lattice30-precise"])   # (precise: no one-launch sweep is built for it -- measured slower; the pair is then twice the same path, still bit-exact with the oracle)
def test_one_launch_sweep_equals_one_launch_per_colour_bit_for_bit(mesh):
    precision = "precise" if mesh.endswith("-precise") else "fast"
    mesh = mesh.split("-")[0]
    if mesh == "lattice30":
        v, t = make_lattice(30, y0=0.01)          # 162,000 tets: 8 colours of ~3,400 clusters
    else:
        v, t = load_mesh(mesh)
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
    a, b = _pair(v, t, precision)
    assert a.info.num_levels == b.info.num_levels >= 2
    total = 0
    for k, (n, dt) in enumerate(((10, DT), (1, DT), (2, DT), (7, DT), (10, DT * 2), (3, DT * 2), (20, DT))):
        if k == 3:
            for body in (a, b):
                body.setGrab(5, [0.05, 0.6, -0.1])
        if k == 5:
            for body in (a, b):
                body.endGrab()
        a.simulateSubsteps(n, dt, PP)               # one graph launch: n sweeps, one kernel each
        for _ in range(n):
            b.simulate(dt, PP)                      # tetsim_step: eager, one kernel per colour
        total += n
        assert _same(a.pos, b.pos) and _same(a.vel, b.vel) and a.volError == b.volError, (mesh, total)
    assert a.pos[:, 1].min() == 0.0                 # floor contact was part of it
    # ... and the other way round: the one-launch body through tetsim_step (its own block of stamps per call), the twin through tetsim_step_n
    blob = a.saveState()
    for _ in range(12):
        a.simulate(DT, PP)
    b.simulateSubsteps(12, DT, PP)
    assert _same(a.pos, b.pos) and _same(a.vel, b.vel)
    a.loadState(blob)
    a.simulateSubsteps(12, DT, PP)
    assert _same(a.pos, b.pos)
    if precision == "precise":      # ... and both are the sequential reference algorithm on the permuted tets
        orc = OracleNH(v, t[a.tetOrder], PP)
        c = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="precise", order="clustered")
        for _ in range(6):
            orc.simulate(DT, PP)
        c.simulateSubsteps(6, DT, PP)
        assert _same(c.pos, orc.pos) and c.volError == orc.volError


def test_one_launch_sweep_stays_inside_the_fast_envelope_against_the_sequential_oracle():
    v, t = load_mesh("dragon")
    body = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order="clustered")
    orc = OracleNH(v, t[body.tetOrder], PP)
    tol = {1: 2e-6, 10: 2e-5, 100: 5e-4}
    for step in range(1, 101):
        body.simulate(DT, PP) if step % 3 else body.simulateSubsteps(1, DT, PP)
        orc.simulate(DT, PP)
        if step in tol:
            within("neo-hookean fast clustered one-launch sweep vs oracle dragon @%d" % step, np.abs(body.pos - orc.pos).max(), tol[step])


def test_one_launch_sweep_long_calls_and_many_calls():
    """Stamps are epoch + substep x colours + colour: a call of more substeps than one block of 65,536 stamps holds is chunked, and
    thousands of calls never reuse a stamp."""
    v, t = make_lattice(6, y0=0.05)
    a, b = _pair(v, t)
    n = 65000 // a.info.num_levels + 50
    a.simulateSubsteps(n, DT, PP)
    b.simulateSubsteps(n, DT, PP)
    assert _same(a.pos, b.pos)
    for _ in range(300):
        a.simulate(DT, PP)
        a.simulateSubsteps(2, DT, PP)
    b.simulateSubsteps(900, DT, PP)
    assert _same(a.pos, b.pos) and np.isfinite(a.pos).all()
