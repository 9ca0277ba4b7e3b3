"""Peer-to-peer halo (tetsim_halo_p2p_export / _connect): the boundary-particle kernel stores the ghost positions straight
into the neighbours' ghost ranges, double buffered by substep parity, and a word per neighbour says "arrived" -- no transfer kernel.
The arithmetic is untouched, so a decomposition stepped with it must equal the same decomposition stepped with the copy transport
BIT FOR BIT -- any number of substeps per call (the buffers alternate by parity), across calls, across dt changes (refresh
exchange), for slabs and for ragged partitions (a boundary particle that two neighbours read)."""
import numpy as np
import pytest

from conftest import load_mesh, within
from tetsim_amd import SoftBodyHIP, TetSimError, comm_info, group_p2p_connect, group_step_n, make_lattice, p2p_connect, p2p_export

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def _parts(v, t, n, owner=None, **kw):
    return [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=n, part_index=p, vert_owner=owner, **kw) for p in range(n)]


def _gather(parts, nv):
    pos = np.empty((nv, 3), np.float32)
    for b in parts:
        pos[b.ownedIds] = b.pos
    return pos


@pytest.mark.parametrize("case", ["slabs2", "slabs5", "dragon3", "dragon3_partitioner"])
def test_p2p_equals_copy_transport_bit_for_bit(case):
    if case.startswith("dragon3"):
        from tetsim_amd.partition import index_range_owner
        v, t = load_mesh("dragon")
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
        # contiguous index ranges of the Dragon: ragged interfaces, two neighbours reading one particle; or the built-in partitioner's cut
        n, owner = 3, (index_range_owner(len(v), 3) if case == "dragon3" else None)
    else:
        n = int(case[5:])
        cells = 16
        v, t = make_lattice(cells, y0=0.02)
        owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * n // (cells + 1), n - 1).astype(np.int32)
    a, b = _parts(v, t, n, owner), _parts(v, t, n, owner)
    group_step_n(a, 3, DT, PP)                               # (a group exists after its first step; connect between steps)
    group_step_n(b, 3, DT, PP)
    group_p2p_connect(b)
    for n_sub, dt in ((20, DT), (1, DT), (7, DT), (2, DT * 2), (5, DT * 2), (20, DT), (3, DT * 0.5)):
        group_step_n(a, n_sub, dt, PP)
        group_step_n(b, n_sub, dt, PP)
        pa, pb = _gather(a, len(v)), _gather(b, len(v))
        assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32)), (case, n_sub, dt)
    for x, y in zip(a, b):
        assert np.array_equal(x.quats.view(np.uint32), y.quats.view(np.uint32)) and np.array_equal(x.vel.view(np.uint32), y.vel.view(np.uint32))
    assert np.isfinite(pb).all() and pb[:, 1].min() < 0.02   # floor contact was part of it
    mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    for n_sub, dt in ((3, DT), (20, DT), (1, DT), (7, DT), (2, DT * 2), (5, DT * 2), (20, DT), (3, DT * 0.5)):
        mono.simulateSubsteps(n_sub, dt, PP)
    within("polar fast p2p group %s vs monolithic @61" % case, np.abs(pb - mono.pos).max(), 2e-4)


def test_p2p_preconditions_and_errors():
    v, t = make_lattice(6, y0=0.3)
    prec = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="precise", part_count=2, part_index=p) for p in range(2)]
    with pytest.raises(TetSimError, match="blocked FAST"):
        p2p_export(prec[0])
    fast = _parts(v, t, 2)
    blobs = [p2p_export(b) for b in fast]
    assert all(len(b) == 512 for b in blobs)
    with pytest.raises(TetSimError, match="one blob per partition"):
        p2p_connect(fast[0], blobs[:1])
    with pytest.raises(TetSimError, match="bad peer blob"):
        p2p_connect(fast[0], [blobs[0], b"\0" * 512])
    # connected without RCCL and without a group, the peer-to-peer halo is the body's only transport (one rank per PROCESS steps that
    # way: test_two_processes_share_one_gpu_through_ipc_mappings) -- and it cannot change dt, which nothing would carry to the ghosts
    for b in fast:
        p2p_connect(b, blobs)
    fast[0].simulate(DT, PP)               # (the first substep after a connection waits for nobody)
    with pytest.raises(TetSimError, match="only transport is the peer-to-peer halo"):
        fast[0].simulate(DT * 2, PP)


def test_p2p_loopback_rank_with_rccl_refresh():
    """One rank whose halo partner is itself (TETSIM_DEBUG_LOOPBACK_HALO; what tools/loopback_rank.py times): the peer-to-peer
    chain -- stores, words of both parities, the captured two-chain graphs incl. odd call lengths, the RCCL refresh after a dt
    change -- runs and stays finite; comm_info says which transport is active."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import numpy as np
from tetsim_amd import SoftBodyHIP, comm_info, comm_init, comm_unique_id, make_lattice, p2p_connect, p2p_export
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -9.0, 2.5, 10.0, 9.0])
v, t = make_lattice(20, nz=60)
owner = np.minimum((np.arange(len(v)) // 21 ** 2) // 20, 2).astype(np.int32)
b = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", part_count=3, part_index=1, vert_owner=owner, ref_fixed_bounds=False)
comm_init(b, comm_unique_id(), 0, 1)
dt = 1 / 1200
b.simulateSubsteps(4, dt, pp)
assert not comm_info(b)["p2p"]
p2p_connect(b, [p2p_export(b)])
assert comm_info(b)["p2p"] and comm_info(b)["loopback"]
for n, d in ((5, dt), (20, dt), (3, dt), (20, dt * 2), (1, dt * 2), (20, dt)):
    b.simulateSubsteps(n, d, pp)
b.sync()
assert np.isfinite(b.pos).all()
print("LOOPBACK_P2P_OK")
"""
    env = dict(os.environ, TETSIM_DEBUG_LOOPBACK_HALO="1", TETSIM_HALO_TIMEOUT_MS="3000", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "LOOPBACK_P2P_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


_RANK_SCRIPT = r"""
import os, sys
import numpy as np
import torch.distributed as dist
from tetsim_amd import SoftBodyHIP, comm_info, make_lattice, p2p_connect, p2p_export
rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
dist.init_process_group(backend="gloo", rank=rank, world_size=world)      # rendezvous only: the halo itself needs no communicator
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
cells = 16
v, t = make_lattice(cells, y0=0.02)
owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * world // (cells + 1), world - 1).astype(np.int32)
body = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", part_count=world, part_index=rank, vert_owner=owner)
blobs = [None] * world
dist.all_gather_object(blobs, p2p_export(body))
p2p_connect(body, blobs)                                                   # other PROCESSES' buffers: opened through HIP IPC handles
dist.barrier()                                                             # nobody steps before everybody is connected
dt = (1.0 / 60.0) / 20
for n in (20, 1, 7, 20, 3):
    body.simulateSubsteps(n, dt, pp)
body.sync()
dist.barrier()
from tetsim_amd import halo_p2p_probe
for _ in range(2):                                                         # the hand-over on this "wire" (one GPU: its own memory), twice: the inbox words count on
    hp = halo_p2p_probe(body, 50)
    assert 0.0 < hp["min"] <= hp["median"] <= hp["max"] < 5000.0, hp
body.simulateSubsteps(4, dt, pp)                                           # ... and the halo's own words are untouched by it
body.sync()
np.save(os.path.join(out, "ids%d.npy" % rank), body.ownedIds)
np.save(os.path.join(out, "pos%d.npy" % rank), body.pos)
dist.barrier()                                                             # keep the mappings alive until every rank has finished stepping
body.close()
print("RANK_OK", rank, flush=True)
"""


def test_two_processes_share_one_gpu_through_ipc_mappings(tmp_path):
    """The multi-process form of the peer-to-peer halo, as far as ONE GPU can show it: two rank processes (gloo for the rendezvous, no
    RCCL communicator at all), each opens the other's ghost buffers and "arrived" words through hipIpcOpenMemHandle, stores into them
    and waits on its own.  Must equal the in-process decomposition stepped with the copy transport bit for bit.  (What a single GPU
    cannot show is the xGMI path between two devices: bench.py's N > 1 run compares the two transports on real peers.)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", TETSIM_HALO_TIMEOUT_MS="5000", PYTHONPATH=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK_OK" in so, (r, so[-300:], se[-1500:])
    cells = 16
    v, t = make_lattice(cells, y0=0.02)
    owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * 2 // (cells + 1), 1).astype(np.int32)
    ref = _parts(v, t, 2, owner)
    for n in (20, 1, 7, 20, 3, 4):
        group_step_n(ref, n, DT, PP)
    want = _gather(ref, len(v))
    got = np.empty_like(want)
    for r in range(2):
        got[np.load(tmp_path / ("ids%d.npy" % r))] = np.load(tmp_path / ("pos%d.npy" % r))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


_CHECKPOINT_SCRIPT = r"""
import os, sys
import numpy as np
import torch.distributed as dist
from tetsim_amd import SoftBodyHIP, make_lattice, p2p_connect, p2p_export
rank, world, out, lean = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1], sys.argv[2] == "lean"
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
cells = 16
v, t = make_lattice(cells, y0=0.02)
owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * world // (cells + 1), world - 1).astype(np.int32)
body = SoftBodyHIP(v, t, None, dict(pp), solver="polar", precision="fast", part_count=world, part_index=rank, vert_owner=owner, lean_state=lean)
blobs = [None] * world
dist.all_gather_object(blobs, p2p_export(body))
p2p_connect(body, blobs)
dist.barrier()
dt = (1.0 / 60.0) / 20
for n in (20, 1):
    body.simulateSubsteps(n, dt, pp)
body.sync()
# 21 substeps: an ODD parity (the ghosts live in the second buffer).  No barrier in front of the save: the neighbour may still be inside
# its last substep -- tetsim_save_state waits for its deliveries itself (and must not touch the live receive buffer of the other parity)
if rank == 1:
    import time; time.sleep(0.2)                  # ... and rank 0 is well into its continuation when rank 1 saves
blob = body.saveState()
for n in (7, 20):
    body.simulateSubsteps(n, dt, pp)
body.sync()
pos_a, quat_a = body.pos, body.quats
dist.barrier()                                     # everybody has finished the continuation
body.loadState(blob)
dist.barrier()                                     # include/tetsim.h: nobody steps before everybody has loaded
for n in (7, 20):
    body.simulateSubsteps(n, dt, pp)
body.sync()
assert np.array_equal(body.pos.view(np.uint32), pos_a.view(np.uint32)), "rank %d: the restored continuation differs" % rank
assert np.array_equal(body.quats.view(np.uint32), quat_a.view(np.uint32))
np.save(os.path.join(out, "ids%d.npy" % rank), body.ownedIds)
np.save(os.path.join(out, "pos%d.npy" % rank), body.pos)
dist.barrier()
body.close()
print("RANK_OK", rank, flush=True)
"""


@pytest.mark.parametrize("mode", ["carried", "lean"])
def test_two_processes_checkpoint_through_ipc_mappings(tmp_path, mode):
    """Checkpoint / resume of one-rank-per-process peer-to-peer bodies (advisor, round 5): quiesce() drains only a rank's own queues
    while the NEIGHBOUR stores into its ghost range, so tetsim_save_state waits for the neighbours' "arrived" words itself, copies the
    odd parity's ghosts straight into the blob (never into the other parity's live receive buffer), and a rank that saves while its
    neighbour already steps on neither reads stale ghosts nor disturbs the running simulation; loads are bracketed by rank barriers
    (include/tetsim.h).  The restored continuation equals the original and the in-process decomposition bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "rank.py"
    script.write_text(_CHECKPOINT_SCRIPT)
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", TETSIM_HALO_TIMEOUT_MS="5000", PYTHONPATH=ROOT,
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, str(script), str(tmp_path), mode], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "RANK_OK" in so, (r, so[-300:], se[-1500:])
    cells = 16
    v, t = make_lattice(cells, y0=0.02)
    owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * 2 // (cells + 1), 1).astype(np.int32)
    ref = _parts(v, t, 2, owner, lean_state=mode == "lean")
    for n in (20, 1, 7, 20):
        group_step_n(ref, n, DT, PP)
    want = _gather(ref, len(v))
    got = np.empty_like(want)
    for r in range(2):
        got[np.load(tmp_path / ("ids%d.npy" % r))] = np.load(tmp_path / ("pos%d.npy" % r))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- two-layer ghost region: ghosts cross only every other substep (TETSIM_FLAG_DEEP_GHOSTS) -------------------------------------------
def _deep_parts(v, t, n, owner):
    return [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=n, part_index=p, vert_owner=owner, deep_ghosts=True) for p in range(n)]


@pytest.mark.parametrize("case", ["slabs2", "slabs4", "dragon3"])
def test_two_layer_ghost_region_tracks_the_monolithic_body_and_its_ghost_tets_track_their_owners(case):
    """The device form of tests/test_partition_gloo.py's algorithm (which is bit-exact there, with the oracle as compute body): a
    partition advances its first ghost layer itself on even substeps, its neighbours' particles arrive after odd ones, the
    second-layer ghost tets are evolved after the fact from the early message.  FAST sums depend on the tiling, which differs from a
    monolithic body's, so positions are compared within the FAST tolerance -- under floor contact and a grab, where a ghost tet
    whose state was one substep stale would show -- and the sharp check is on the STATE: after an even number of substeps every
    ghost tet's quaternion (second layer included) equals the copy its owner carries to rounding; a missed late evolve leaves it a
    whole substep's rotation behind."""
    if case == "dragon3":
        v, t = load_mesh("dragon")
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
        n, owner = 3, None
        # (a gentle pull: yanking a Dragon vertex by a third of the body's size folds its tets over, the polar iteration settles on
        # another branch from a rounding's difference, and ANY decomposition -- one ghost layer too -- is millimetres off the
        # monolithic body at once: tools/attic/deep_diag.py, profiles/archive/r03_loopback.txt section 3)
        pull = (0.02, 0.04)
    else:
        n = int(case[5:])
        cells = 16
        v, t = make_lattice(cells, y0=0.02)
        owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * n // (cells + 1), n - 1).astype(np.int32)
        pull = (0.15, 0.3)
    parts = _deep_parts(v, t, n, owner)
    assert all(p.info.local_elems > q.info.local_elems for p, q in zip(parts, _parts(v, t, n, owner)))   # the second layer of ghost tets is there
    with pytest.raises(TetSimError):                                       # such a body steps through the peer-to-peer halo only
        group_step_n(parts, 1, DT, PP)
    group_p2p_connect(parts)
    mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
    gid = int(parts[0].ownedIds[0])
    total = 0
    for k, n_sub in enumerate((20, 1, 7, 2, 20, 3, 5, 20)):              # odd call lengths: an exchange may straddle two calls
        if k == 2:
            for b in parts + [mono]:
                b.setGrab(gid, [float(v[gid, 0]) + pull[0], float(v[gid, 1]) + pull[1], float(v[gid, 2])])
        if k == 6:
            for b in parts + [mono]:
                b.endGrab()
        group_step_n(parts, n_sub, DT, PP)
        mono.simulateSubsteps(n_sub, DT, PP)
        total += n_sub
    pos = _gather(parts, len(v))
    assert np.isfinite(pos).all()
    # (the grab is a violent transient: a one-layer decomposition is 3e-4 off the monolithic body ten substeps into it, too;
    # in calm phases the redundant first ghost layer costs ~2e-7 m per substep -- the twin copies of a straddling tet see inputs that
    # differ by a rounding on odd substeps)
    within("polar fast two-layer ghosts %s vs monolithic @%d" % (case, total), np.abs(pos - mono.pos).max(), 2e-3)
    # ghost tets against their owners' copies (total is even: the last exchange has happened)
    assert total % 2 == 0
    owner_q = {}
    for b in parts:
        for gt, q in zip(b.localTets[:b.info.owned_elems] if False else b.localTets, b.quats):
            owner_q.setdefault(int(gt), []).append(q)
    worst = max(float(min(np.abs(np.array(qs) - qs[0]).max(), np.abs(np.array(qs[1:]) + qs[0]).max())) for qs in owner_q.values() if len(qs) > 1)   # (q and -q are one rotation)
    assert sum(len(qs) > 1 for qs in owner_q.values()) > 50
    within("polar fast two-layer ghosts %s ghost-tet quaternions vs owners @%d" % (case, total), worst, 2e-4)
    with pytest.raises(TetSimError):                                       # ... and cannot change dt
        group_step_n(parts, 1, DT * 2, PP)
