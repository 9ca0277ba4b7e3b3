"""GPU parity, Neo-Hookean XPBD Gauss-Seidel (BASELINE configs 1 and 4), through the C ABI.

PRECISE mode is compared BIT FOR BIT with (a) golden vectors recorded from the reference's Softbody.js and
(b) the CPU oracle (itself pinned to those goldens) on meshes the goldens do not cover.
FAST mode (f32 + FMA) is compared at the tolerances stated in the tests.
"""
import numpy as np
import pytest

from conftest import case_dt, load_f32, load_mesh, sha16, within
from oracle import OracleNH
from tetsim_amd import SoftBodyHIP, make_lattice

pytestmark = pytest.mark.gpu

GOLD_CASES = ["dragon", "dragon_sub5", "dragon_grab", "dragon_soft", "lat4", "lat4c", "lat2degen", "notets"]


def _run_case(body, c, nsteps, on_step):
    dt = case_dt(c)
    for step in range(1, nsteps + 1):
        for ev in c["grab"]:
            if ev["at"] != step:
                continue
            if ev["op"] == "start":
                body.startGrab(ev["p"])
            elif ev["op"] == "move":
                body.moveGrabbed(ev["p"])
            else:
                body.endGrab()
        body.simulate(dt, c["params"])
        on_step(step)


@pytest.mark.parametrize("name", GOLD_CASES)
def test_precise_bit_exact_vs_reference_goldens(name, golden):
    gold, cases = golden
    c, g = cases[name], gold[name]
    nsteps = min(c["nsteps"], 200)
    v, t = load_mesh(c["mesh"])
    body = SoftBodyHIP(v, t, None, c["params"], solver="neohookean", precision="precise", order="original")
    grab_ids = []

    def on_step(step):
        if body.grabId >= 0 and (not grab_ids or grab_ids[-1] != body.grabId):
            grab_ids.append(body.grabId)
        gs = g["steps"].get(str(step))
        if gs:
            assert sha16(body.pos) == gs["pos"], (name, step)
            assert sha16(body.vel) == gs["vel"], (name, step)
            assert sha16(body.prevPos) == gs["prev"], (name, step)
            if gs["volError"] is not None:
                assert body.volError == gs["volError"], (name, step)
        if step in c["dumps"]:
            assert np.array_equal(body.pos.ravel().view(np.uint32), load_f32(f"{name}_pos_{step}.f32").view(np.uint32))

    _run_case(body, c, nsteps, on_step)
    if c["grab"]:
        assert grab_ids == g["grabIds"]
    assert np.array_equal(body.invMass.view(np.uint32), load_f32(c["mesh"] + "_invMass.f32").view(np.uint32))


def test_dragon_1200_substeps_hash(golden):
    """SURVEY.md §8(c) known answer: two simulated seconds including floor contact, via graph launches."""
    gold, cases = golden
    c, g = cases["dragon"], gold["dragon"]
    v, t = load_mesh("dragon")
    body = SoftBodyHIP(v, t, None, c["params"], solver="neohookean", precision="precise")
    dt = case_dt(c)
    for frame in range(120):
        body.simulateSubsteps(10, dt, c["params"])
        if frame == 59:
            assert sha16(body.pos) == g["steps"]["600"]["pos"]
    assert sha16(body.pos) == g["steps"]["1200"]["pos"] == "9f52c76cdd7223f0"


@pytest.mark.parametrize("order", ["original", "coloured", "clustered"])
def test_precise_bit_exact_vs_oracle_lattice(order):
    """Config 4's claim: the coloured schedule equals the sequential solver fed the permuted tetIds."""
    v, t = make_lattice(6, y0=0.02)
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
              worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
    body = SoftBodyHIP(v, t, None, pp, solver="neohookean", precision="precise", order=order)
    perm = body.tetOrder
    assert sorted(perm.tolist()) == list(range(len(t)))
    if order == "original":
        assert np.array_equal(perm, np.arange(len(t)))
    lo = body.levelOffsets
    assert lo[0] == 0 and lo[-1] == len(t) and np.all(np.diff(lo) > 0)
    orc = OracleNH(v, t[perm], pp)
    dt = (1.0 / 60.0) / 10
    for step in range(60):
        body.simulate(dt, pp)
        orc.simulate(dt, pp)
        if step in (0, 9, 59):
            assert np.array_equal(body.pos.view(np.uint32), orc.pos.view(np.uint32)), step
            assert np.array_equal(body.vel.view(np.uint32), orc.vel.view(np.uint32)), step
            assert body.volError == orc.volError
    if order == "coloured":
        assert body.info.num_levels <= 64  # a colouring, not a wavefront
    if order == "clustered":
        assert body.info.num_levels == 8   # cells as clusters, 2x2x2 cell parities as cluster colours


@pytest.mark.parametrize("precision", ["precise", "fast"])
def test_clustered_dragon_vs_oracle(precision):
    """The clustered schedule on the reference's own mesh: bit-exact (PRECISE) with the sequential solver fed the
    permuted tetIds, FAST within the Neo-Hookean FAST tolerance of the same horizon; grab + floor included."""
    from conftest import load_mesh
    v, t = load_mesh("dragon")
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
              worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
    body = SoftBodyHIP(v, t, None, pp, solver="neohookean", precision=precision, order="clustered")
    assert body.info.num_levels < 32        # fewer launches than tet colours
    orc = OracleNH(v, t[body.tetOrder], pp)
    dt = (1.0 / 60.0) / 10
    for frame in range(10):
        if frame == 3:
            body.setGrab(100, [0.3, 1.5, 0.1])
            orc.setGrab(100, [0.3, 1.5, 0.1])
        if frame == 6:
            body.endGrab()
            orc.endGrab()
        body.simulateSubsteps(10, dt, pp)
        for _ in range(10):
            orc.simulate(dt, pp)
        if precision == "precise":
            assert np.array_equal(body.pos.view(np.uint32), orc.pos.view(np.uint32)), frame
            assert body.volError == orc.volError
        else:
            within("neo-hookean fast clustered dragon vs oracle frame %d" % frame, np.abs(body.pos - orc.pos).max(), 2e-3)
            # (the four-lane kernel's det F - 1, stored by the x lane of every quad)
            within("neo-hookean fast clustered dragon volError vs oracle frame %d" % frame, abs(body.volError - orc.volError), 5e-6)


def test_fast_tolerance_vs_reference_goldens(golden):
    """f32 + FMA kernel vs the reference's f64-compute trajectory.  The system is stiff (|v| reaches 6 m/s after
    one substep), so the tolerance is stated per horizon: max |dx| over all particles."""
    gold, cases = golden
    c = cases["dragon"]
    v, t = load_mesh("dragon")
    body = SoftBodyHIP(v, t, None, c["params"], solver="neohookean", precision="fast")
    dt = case_dt(c)
    tol = {1: 2e-6, 10: 2e-5, 100: 2e-3}
    for step in range(1, 101):
        body.simulate(dt, c["params"])
        if step in tol:
            ref = load_f32(f"dragon_pos_{step}.f32").reshape(-1, 3)
            within("neo-hookean fast dragon vs reference golden @%d" % step, np.abs(body.pos - ref).max(), tol[step])


@pytest.mark.parametrize("precision", ["precise", "fast"])
@pytest.mark.parametrize("order", ["coloured", "clustered"])
def test_graph_equals_eager(order, precision):
    """tetsim_step_n fuses the particle pass that ends a substep with the prediction that starts the next; clustered schedules fold
    it into the sweep itself -- the lane of the first cluster to touch a particle does it while loading the particle
    (nh_kernels.inc: fold_particle), particles no cluster touches keep a small pass of their own.  Same operations per particle:
    n substeps in one call equal n tetsim_step calls bit for bit, in both arithmetics, with floor contact, a grab, and two
    particles that belong to no tet."""
    v, t = make_lattice(5, y0=0.01)
    v = np.vstack([v, np.float32([[0.3, 0.02, 0.2], [-0.4, 1.5, 0.1]])])      # two loose particles: free fall, one hits the floor
    pp = dict(density=1000.0, friction=100.0)
    a = SoftBodyHIP(v, t, None, pp, solver="neohookean", order=order, precision=precision)
    b = SoftBodyHIP(v, t, None, pp, solver="neohookean", order=order, precision=precision)
    dt = 1.0 / 600
    for k, n in enumerate((7, 1, 12, 2)):
        if k == 1:
            for body in (a, b):
                body.setGrab(40, [0.2, 0.8, -0.1])
        if k == 3:
            for body in (a, b):
                body.endGrab()
        a.simulateSubsteps(n, dt, pp)
        for _ in range(n):
            b.simulate(dt, pp)
        assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32)), (k, n)
        assert np.array_equal(a.vel.view(np.uint32), b.vel.view(np.uint32)) and np.array_equal(a.prevPos.view(np.uint32), b.prevPos.view(np.uint32))
    assert a.pos[:, 1].min() == 0.0 and a.pos[-2, 1] < 0.02


def test_repeated_vertex_is_rejected():
    from tetsim_amd import TetSimError
    v, t = make_lattice(2)
    t = t.copy()
    t[3, 1] = t[3, 0]
    with pytest.raises(TetSimError):
        SoftBodyHIP(v, t, None, {}, solver="neohookean")


@pytest.mark.parametrize("order", ["coloured", "clustered"])
def test_lattice_1m_coloured_properties(order):
    """BASELINE config 4 at full size (998,250 tets), through size-independent properties: determinism, PRECISE vs FAST
    agreement, rigid free fall of the centroid (XPBD constraints are internal forces), bounded volume error."""
    v, t = make_lattice(55)
    pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
              worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
    dt, n = (1.0 / 60.0) / 20, 20
    a = SoftBodyHIP(v, t, None, pp, solver="neohookean", precision="precise", order=order)
    b = SoftBodyHIP(v, t, None, pp, solver="neohookean", precision="precise", order=order)
    f = SoftBodyHIP(v, t, None, pp, solver="neohookean", precision="fast", order=order)
    for body in (a, b, f):
        body.simulateSubsteps(n, dt, pp)
    pa, pb, pf = a.pos, b.pos, f.pos
    assert a.info.num_levels <= (40 if order == "coloured" else 8)  # a colouring (max valence 24), not a wavefront
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))   # deterministic
    assert np.isfinite(pa).all()
    within("neo-hookean fast vs precise %s 1M lattice @20" % order, np.abs(pa - pf).max(), 2e-4)   # f32+FMA tracks the f64-exact path
    # the masses are lumped per vertex, so the MASS-weighted centroid falls rigidly: sum_k (k dt) dt g = g dt^2 n(n+1)/2
    m = 1.0 / a.invMass.astype(np.float64)
    drop = ((pa[:, 1].astype(np.float64) - v[:, 1]) * m).sum() / m.sum()
    assert abs(drop - (-9.81 * dt * dt * n * (n + 1) / 2)) < 1e-5   # 1.43 mm expected; f32 position stores cost ~4e-6
    assert abs(((pa[:, 0].astype(np.float64) - v[:, 0]) * m).sum() / m.sum()) < 1e-6
    assert abs(a.volError) < 0.2 and a.volError == b.volError
