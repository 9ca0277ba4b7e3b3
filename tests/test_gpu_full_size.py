"""Oracle parity at BASELINE.json's FULL sizes (configs 3, 4, 5), through the C ABI.

The CPU restatement handles these sizes in seconds once it may use the host's cores (OpenMP over tets / particles for the
polar solver; the Neo-Hookean sweep is sequential by definition, 0.1 s per substep), so the 1 M-tet and 8 M-tet bodies are
compared with it directly -- the blocked kernel's 3,900 / 31,000 workgroup tiles, its ELL partial-sum columns (up to 9 per
particle) and the cluster / colour schedules of a million tets all sit under an oracle here, not only under invariants.
Tolerances: absolute position error in metres, <= 3 x the value observed on MI355X (conftest.within); the Neo-Hookean
PRECISE schedules are bit-exact.
"""
import numpy as np
import pytest

from conftest import within
from oracle import OracleNH, OraclePJ, set_threads
from tetsim_amd import SoftBodyHIP, make_lattice

pytestmark = pytest.mark.gpu

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT20 = (1.0 * (1.0 / 60.0)) / 20


@pytest.fixture(scope="module")
def lattice_1m():
    v, t = make_lattice(55)
    assert len(t) == 998250 and len(v) == 175616
    return v, t


@pytest.fixture(scope="module")
def oracle_1m_polar(lattice_1m):
    """OraclePJ trajectories of the 1 M-tet lattice: {drop height: {substep: positions}} (16 OpenMP threads; the oracle's
    result does not depend on the thread count: every tet and every particle is computed independently, in slot order)."""
    v, t = lattice_1m
    out = {}
    set_threads(16)
    try:
        for y0, steps, dumps in ((0.5, 20, (1, 20)), (0.0005, 20, (20,))):
            vv = v.copy()
            vv[:, 1] += np.float32(y0 - 0.5)
            o = OraclePJ(vv, t, PP, slot_quirk=True)
            got = {}
            for s in range(1, steps + 1):
                o.simulate(DT20, PP)
                if s in dumps:
                    got[s] = (o.pos, o.quats)
            out[y0] = (vv, got)
            del o
    finally:
        set_threads(1)
    return out


@pytest.mark.parametrize("mode,tol1,tol20", [("precise", 2.5e-7, 1e-6), ("fast", 4e-6, 2e-4), ("fast-gather", 4e-6, 5e-5), ("fast-lean", 4e-6, 2e-4)])
def test_lattice_1m_polar_vs_oracle(mode, tol1, tol20, lattice_1m, oracle_1m_polar):
    """BASELINE config 3: free fall from 0.5 m (substeps 1 and 20), through tetsim_step and tetsim_step_n."""
    _, t = lattice_1m
    vv, ref = oracle_1m_polar[0.5]
    kw = dict(precision="precise") if mode == "precise" else dict(precision="fast", gather=mode == "fast-gather", lean_state=mode == "fast-lean")
    body = SoftBodyHIP(vv, t, None, dict(PP), solver="polar", **kw)
    body.simulate(DT20, PP)
    within("polar %s 1M lattice vs oracle @1" % mode, np.abs(body.pos - ref[1][0]).max(), tol1)
    body.simulateSubsteps(19, DT20, PP)
    within("polar %s 1M lattice vs oracle @20" % mode, np.abs(body.pos - ref[20][0]).max(), tol20)
    q = body.quats
    qq = np.empty_like(q)
    qq[body.localTets] = q
    within("polar %s 1M lattice vs oracle @20 (quat)" % mode, np.abs(qq - ref[20][1]).max(), 2.5e-7 if mode == "precise" else 1e-3)
    assert np.abs(np.linalg.norm(q, axis=1) - 1.0).max() < 1e-5


@pytest.mark.parametrize("mode,tol", [("precise", 1e-6), ("fast", 1e-4), ("fast-lean", 1e-4)])
def test_lattice_1m_polar_floor_contact_vs_oracle(mode, tol, lattice_1m, oracle_1m_polar):
    """The same body dropped 0.5 mm onto the floor: 20 substeps with the bottom face in contact (clamp + friction branch of
    the particle pass on 3,136 particles, deformation in the tiles above it)."""
    _, t = lattice_1m
    vv, ref = oracle_1m_polar[0.0005]
    body = SoftBodyHIP(vv, t, None, dict(PP), solver="polar", precision=mode.split("-")[0], lean_state=mode == "fast-lean")
    body.simulateSubsteps(20, DT20, PP)
    p = body.pos
    assert p[:, 1].min() == 0.0 and ref[20][0][:, 1].min() == 0.0
    within("polar %s 1M lattice floor contact vs oracle @20" % mode, np.abs(p - ref[20][0]).max(), tol)


@pytest.mark.parametrize("order", ["coloured", "clustered"])
def test_lattice_1m_neohookean_bit_exact(order, lattice_1m):
    """BASELINE config 4: the parallel Gauss-Seidel schedules of a million tets against the SEQUENTIAL reference algorithm
    (Softbody.js:207-208) fed tetIds[tetsim_get_tet_order()]: positions, velocities and volError bit for bit, 5 substeps."""
    v, t = lattice_1m
    vv = v.copy()
    vv[:, 1] -= np.float32(0.4995)   # 0.5 mm above the floor: contact from the second substep on
    body = SoftBodyHIP(vv, t, None, dict(PP), solver="neohookean", precision="precise", order=order)
    orc = OracleNH(vv, t[body.tetOrder], PP)
    assert np.array_equal(body.invMass.view(np.uint32), orc.invMass.view(np.uint32))
    for s in range(5):
        body.simulate(DT20, PP)
        orc.simulate(DT20, PP)
        if s in (0, 4):
            assert np.array_equal(body.pos.view(np.uint32), orc.pos.view(np.uint32)), (order, s)
            assert body.volError == orc.volError, (order, s)
    assert np.array_equal(body.vel.view(np.uint32), orc.vel.view(np.uint32))
    assert body.pos[:, 1].min() == 0.0
    # FAST (f32 + FMA + v_rcp / v_rsq) against the same oracle, in free fall: the floor branch (`y < 0` -> clamp + friction jump,
    # Softbody.js:218-226) is a discontinuity that turns a last-ulp difference into a 1e-4 m one and would measure nothing
    del orc
    fast = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", precision="fast", order=order)
    orc = OracleNH(v, t[fast.tetOrder], PP)
    fast.simulateSubsteps(5, DT20, PP)
    for _ in range(5):
        orc.simulate(DT20, PP)
    within("neo-hookean fast %s 1M lattice vs oracle @5 (free fall)" % order, np.abs(fast.pos - orc.pos).max(), 2e-5)


def test_lattice_8m_polar_vs_oracle():
    """BASELINE config 5's body (110^3 cells, 7,986,000 tets) as ONE handle against the oracle, 2 substeps."""
    v, t = make_lattice(110)
    assert len(t) == 7986000
    set_threads(16)
    try:
        o = OraclePJ(v, t, PP, slot_quirk=True)
        o.simulate(DT20, PP)
        o.simulate(DT20, PP)
        ref = o.pos
        del o
    finally:
        set_threads(1)
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    body.simulateSubsteps(2, DT20, PP)
    within("polar fast 8M lattice vs oracle @2", np.abs(body.pos - ref).max(), 1e-5)
