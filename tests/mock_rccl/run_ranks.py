"""Drives the RCCL-mode halo path of libtetsim_hip with N ranks living in one process (one host thread per rank, all on
device 0) against the strict test double of tests/mock_rccl/mock_rccl.cpp.  Run with TETSIM_RCCL_LIB pointing at the double.

    python tests/mock_rccl/run_ranks.py <nranks> <precise|fast> <cells> <substep calls> <substeps per call>
Prints "OK max|dx| = ..." and exits 0 when the stitched result equals the monolithic body (bit for bit in PRECISE)."""
import os
import sys
import threading

import numpy as np

os.environ["TETSIM_HALO_GRAPH"] = "0"   # the test double rendezvouses on the host: not capturable (the real RCCL is: tools/loopback_rank.py)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tetsim_amd import SoftBodyHIP, comm_init, comm_unique_id, make_lattice  # noqa: E402

nranks, precision, cells, calls, per_call = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
assert "mock_rccl" in os.environ.get("TETSIM_RCCL_LIB", ""), "this driver is for the test double only"
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20
v, t = make_lattice(cells, nz=cells * nranks, y0=0.02)          # close to the floor: contact within the run
plane = (cells + 1) * (cells + 1)
owner = np.minimum((np.arange(len(v)) // plane) // cells, nranks - 1).astype(np.int32)
lean = precision == "fast-lean"          # TETSIM_FLAG_LEAN_STATE over the RCCL halo: ghost tets evolve their three corners identically on both sides of a cut
precision = precision.split("-")[0]
kw = dict(solver="polar", precision=precision, ref_fixed_bounds=False, lean_state=lean)

dts = [DT * (2.0 if c == 2 else 0.5 if c == 4 else 1.0) for c in range(calls)]   # the time step changes twice mid-run
# an embedded visual mesh over the whole body: every rank skins the rows whose tet it owns (ghost corners fetched over the transport)
rng = np.random.default_rng(11)
w = rng.dirichlet(np.ones(4), size=5000).astype(np.float32)
vis = np.concatenate([rng.integers(0, len(t), size=(5000, 1)).astype(np.float32), w[:, :3]], axis=1)
mono = SoftBodyHIP(v, t, None, dict(PP), vis, **kw)
for c in range(calls):
    mono.simulateSubsteps(per_call, dts[c], PP)
want, want_vis = mono.pos, mono.visualPositions()

uid = comm_unique_id()
results, errors = [None] * nranks, []


def rank_main(r):
    try:
        body = SoftBodyHIP(v, t, None, dict(PP), vis, part_count=nranks, part_index=r, vert_owner=owner, **kw)
        comm_init(body, uid, r, nranks)                       # blocks until every rank joined (like ncclCommInitRank)
        for c in range(calls):
            if c % 2:
                for _ in range(per_call):
                    body.simulate(dts[c], PP)                    # tetsim_step, one substep per call
            else:
                body.simulateSubsteps(per_call, dts[c], PP)      # tetsim_step_n
        body.refreshFinalGhosts()                                # a collective of the ranks, once per frame; the read below never communicates
        results[r] = (body.ownedIds, body.pos, body.visualIds, body.visualPositions())
        # checkpoint / resume over the RCCL transport: every rank saves its blob, steps on, goes back, steps again -- the same bits
        blob = body.saveState()
        body.simulateSubsteps(5, dts[-1], PP)
        after = body.pos.copy()
        body.loadState(blob)
        body.simulateSubsteps(5, dts[-1], PP)
        assert np.array_equal(body.pos.view(np.uint32), after.view(np.uint32)), "rank %d: the restored partition left the trajectory" % r
        if precision == "fast" and cells >= 20:   # (thinner slabs: some ranks have no interior tiles and refuse, others would wait)
            # tetsim_profile on a body with an RCCL halo: every rank together, interior tet kernel timed
            try:
                pr = body.profile(6, DT, PP)
            except Exception as e:  # noqa: BLE001 -- a slab this thin has no interior tiles: a clean refusal, on every rank alike
                assert "no interior tiles" in str(e), e
                pr = None
            if pr is not None:
                assert pr["substeps"] == 6 and pr["tet_launches"] == 6 and 0 < pr["tets_per_tet_launch"] <= body.info.local_elems, pr
                assert 0.0 < pr["tet_ms"] and 0.0 < pr["vertex_ms"], pr
        body.close()
    except Exception as e:  # noqa: BLE001
        errors.append("rank %d: %r" % (r, e))


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(nranks)]
for th in threads:
    th.start()
for th in threads:
    th.join(timeout=240)
if errors or any(th.is_alive() for th in threads):
    print("FAILED", errors, [th.is_alive() for th in threads], flush=True)
    os._exit(1)
got, got_vis = np.full_like(want, np.nan), np.full_like(want_vis, np.nan)
for ids, pos, vids, vpos in results:
    got[ids] = pos
    got_vis[vids] = vpos
err = max(float(np.abs(got - want).max()), float(np.abs(got_vis - want_vis).max()))
exact = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(got_vis.view(np.uint32), want_vis.view(np.uint32))
ok = exact if precision == "precise" else err <= 1e-4   # FAST: tile composition differs between decompositions (summation order)
print("%s max|dx| = %.3g (bit-exact: %s), ymin %.4f" % ("OK" if ok else "MISMATCH", err, exact, float(want[:, 1].min())))
sys.exit(0 if ok else 2)
