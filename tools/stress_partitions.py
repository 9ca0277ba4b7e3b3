"""Randomised stress of the partitioned (multi-GPU) path on ONE GPU, for minutes: random meshes (lattices of random shape, the Dragon through
the library's own partitioner), 2..8 partitions, the copy transport / the peer-to-peer halo (connected at a random moment) / two-layer ghost
regions, calls of random length, a grab that moves, checkpoints saved and restored at random moments into a freshly built group -- against
the unpartitioned body fed the same calls.  PRECISE: bit for bit.  FAST: finite and within 1e-2 m (two valid FAST trajectories -- different summation orders across the cut -- drift apart
at rounding level and contact / a particle dragged 2 cm per call amplify it over up to 260 substeps: 2-4e-4 m is usual, up to 5e-3 m in the tail
of trials with a drag, on every transport alike; the deviation grows smoothly and alike for every transport, profiles/r05_partition_stress.txt; a lost or stale ghost is centimetres).  python tools/stress_partitions.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_mesh
from tetsim_amd import SoftBodyHIP, group_p2p_connect, group_step_n, make_lattice

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ONLY = os.environ.get("STRESS_TRANSPORT")      # development: one transport, FAST, always a drag -- the same trials for every value (same seed)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-20.0, -1.0, -20.0, 20.0, 30.0, 20.0])   # (ref_fixed_bounds=False below: long lattices must not be squashed by the reference's hard-coded +-2.5 m)
DT = (1.0 / 60.0) / 20
same = lambda a, b: np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
dragon = load_mesh("dragon")
t0 = time.time(); trials = bad = substeps = 0
worst = 0.0
errs = []
kinds = {}
while time.time() - t0 < budget:
    trials += 1
    precision = "precise" if rng.random() < 0.4 else "fast"
    if ONLY:
        precision = "fast"
    if rng.random() < 0.2:
        v, t = dragon
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
        parts, owner, what = int(rng.integers(2, 6)), None, "dragon"
    else:
        n, nz = int(rng.integers(4, 15)), int(rng.integers(8, 40))
        v, t = make_lattice(n, nz=nz, y0=float(rng.choice([0.01, 0.05, 0.4])))
        parts = int(rng.integers(2, min(8, nz // 2) + 1))
        plane = (n + 1) * (n + 1)
        owner = np.minimum((np.arange(len(v)) // plane) * parts // (nz + 1), parts - 1).astype(np.int32)
        what = "lattice %dx%dx%d" % (n, n, nz)
    # (the peer-to-peer halo and two-layer ghost regions belong to the blocked FAST formulation)
    transport = str(rng.choice((["copy", "p2p", "deep"] if owner is not None else ["copy", "p2p"]) if precision == "fast" else ["copy"]))
    if ONLY:
        if owner is None:
            continue
        transport = ONLY
    kw = dict(ref_fixed_bounds=False, **(dict(deep_ghosts=True) if transport == "deep" else {}))
    mk = lambda: [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, part_count=parts, part_index=p, vert_owner=owner, **kw) for p in range(parts)]
    try:
        g = mk()
    except Exception as e:   # (a cut the two-layer ghost region cannot serve: slabs too thin -- the library says so)
        if transport == "deep" and ("layer" in str(e) or "DEEP" in str(e)):
            continue
        raise
    mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, ref_fixed_bounds=False)
    connect_at = int(rng.integers(0, 3)) if transport != "copy" else -1
    if transport == "deep":
        connect_at = 0          # (two-layer ghost regions step through the peer-to-peer halo only)
    calls = int(rng.integers(3, 9))
    pp = dict(PP)
    blobs = None
    msg = None
    grab = int(rng.integers(0, len(v))) if (rng.random() < 0.3 or ONLY) else -1
    for c in range(calls):
        if c == connect_at:
            group_p2p_connect(g)
        n = int(rng.choice([1, 2, 3, 7, 20, 33]))
        if grab >= 0:
            gp = (v[grab] + np.float32([0.02 * c, 0.01, 0.0])).tolist()
            mono.setGrab(grab, gp)
            for b in g:
                b.setGrab(grab, gp)
        mono.simulateSubsteps(n, DT, pp)
        group_step_n(g, n, DT, pp)
        substeps += n
        if transport != "deep" and blobs is None and rng.random() < 0.3:      # checkpoint now, restored into a new group two calls later
            blobs, mblob, at = [b.saveState() for b in g], mono.saveState(), c
        elif blobs is not None and c >= at + 2:
            g2 = mk()
            if connect_at >= 0:
                group_p2p_connect(g2)
            for b, x in zip(g2, blobs):
                b.loadState(x)
            mono.loadState(mblob)
            for b in g:
                b.close()
            g, blobs = g2, None
            if grab >= 0:
                for b in g:
                    b.setGrab(grab, gp)
    pos = np.empty((len(v), 3), np.float32)
    for b in g:
        pos[b.ownedIds] = b.pos
    ref = mono.pos
    err = float(np.abs(pos - ref).max())
    ok = same(pos, ref) if precision == "precise" else (np.isfinite(pos).all() and err < 1e-2)
    if precision == "fast":
        worst = max(worst, err)
        errs.append(err)
    kinds[(transport, precision)] = kinds.get((transport, precision), 0) + 1
    if not ok:
        bad += 1
        print("FAIL trial %d: %s, %d parts, %s, %s, calls %d, grab %d: max |dx| %.3g" % (trials, what, parts, transport, precision, calls, grab, err), flush=True)
    for b in g:
        b.close()
    mono.close()
print("partition stress: %d trials (%s), %d partitioned substeps in %.0f s: %d failures"
      % (trials, ", ".join("%s/%s %d" % (k[0], k[1], n) for k, n in sorted(kinds.items())), substeps, time.time() - t0, bad))
print("largest FAST deviation from the unpartitioned body: %.3g m" % worst)
if ONLY and errs:
    q = np.quantile(np.array(errs), [0.5, 0.9, 0.99, 1.0])
    print("transport %s: %d FAST trials with a drag, deviation median %.2e  p90 %.2e  p99 %.2e  max %.2e" % (ONLY, len(errs), q[0], q[1], q[2], q[3]))
sys.exit(1 if bad else 0)
