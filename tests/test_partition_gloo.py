"""The N>1 path on CPU: world_size-2/3 `gloo` processes run the partition plan of the PRODUCT (tetsim_plan_*,
the routine tetsim_create uses) with the CPU oracle as the per-partition compute body, exchange ghost state
over torch.distributed exactly where the device path issues ncclSend/ncclRecv, and must reproduce the
single-process run bit for bit (Jacobi; global slot order is kept)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_mesh
from tetsim_amd import make_lattice
from tetsim_amd.partition import PartitionPlan, index_range_owner, slab_owner

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def _mesh(kind):
    if kind == "slab":
        v, t = make_lattice(4, nz=9, y0=0.03)
        return v, t, lambda world: slab_owner(4, 9, world)
    v, t = load_mesh("dragon")
    if kind == "dragon_ranges":
        return v, t, lambda world: index_range_owner(len(v), world)  # the file's vertex order cut into ranges: ragged, non-contiguous halos
    return v, t, lambda world: None  # no owner given: the built-in partitioner (tetsim_prep_partition without coordinates)


def _worker(rank, world, port, kind, nsteps, out_dir):
    from oracle import OraclePJ
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v, t, owner_fn = _mesh(kind)
    plan = PartitionPlan(t, len(v), world, rank, owner_fn(world))
    lv = v[plan.local_to_global_vert]
    quirk = plan.n_local_tets > 0 and plan.local_to_global_tet[0] == 0  # SoftbodyGPU.js:568 concerns GLOBAL tet 0
    body = OraclePJ(lv, plan.local_tets, PP, slot_quirk=bool(quirk))
    for _ in range(nsteps):
        body.simulate(DT, PP)
        pos, vel = body.pos, body.vel
        reqs, recv = [], []
        for nb in plan.neighbours:  # same pairing as halo_rccl(): one send + one recv per neighbour
            buf = torch.from_numpy(np.concatenate([pos[nb.send_local], vel[nb.send_local]], axis=1).copy())
            reqs.append(dist.isend(buf, nb.rank))
            r = torch.empty((nb.recv_count, 6), dtype=torch.float32)
            reqs.append(dist.irecv(r, nb.rank))
            recv.append((nb, r))
        for q in reqs:
            q.wait()
        for nb, r in recv:
            a = r.numpy()
            body.writeParticles(np.arange(nb.recv_start, nb.recv_start + nb.recv_count), a[:, :3], a[:, 3:])
    np.save(os.path.join(out_dir, "pos%d.npy" % rank), body.pos[:plan.n_owned])
    np.save(os.path.join(out_dir, "ids%d.npy" % rank), plan.local_to_global_vert[:plan.n_owned])
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    # (below the kernel's ephemeral range: a port that bind(0) hands out may be taken by any outgoing connection before the workers bind it)
    import random
    rng = random.Random(os.getpid() ^ int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(20000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
                return port
            except OSError:
                continue
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,world", [("slab", 2), ("slab", 3), ("dragon", 2), ("dragon", 3), ("dragon_ranges", 2)])
def test_partitioned_gloo_equals_single_process(kind, world, tmp_path):
    from oracle import OraclePJ
    nsteps = 30
    mp.spawn(_worker, args=(world, _free_port(), kind, nsteps, str(tmp_path)), nprocs=world, join=True)
    v, t, _ = _mesh(kind)
    mono = OraclePJ(v, t, PP, slot_quirk=True)
    for _ in range(nsteps):
        mono.simulate(DT, PP)
    ref = mono.pos
    seen = np.zeros(len(v), dtype=bool)
    for r in range(world):
        pos, ids = np.load(tmp_path / ("pos%d.npy" % r)), np.load(tmp_path / ("ids%d.npy" % r))
        assert not seen[ids].any()
        seen[ids] = True
        assert np.array_equal(pos.view(np.uint32), ref[ids].view(np.uint32)), (kind, world, r)
    assert seen.all()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_plan_is_symmetric_and_complete(world):
    v, t = make_lattice(4, nz=16)
    owner = slab_owner(4, 16, world)
    plans = [PartitionPlan(t, len(v), world, r, owner) for r in range(world)]
    assert sum(p.n_owned for p in plans) == len(v)
    assert sum(p.n_owned_tets for p in plans) == len(t)
    for p in plans:
        g = p.local_to_global_vert
        assert np.all(owner[g[:p.n_owned]] == p.part_index) and np.all(owner[g[p.n_owned:]] != p.part_index)
        assert np.array_equal(g[p.local_tets], t[p.local_to_global_tet])          # local connectivity is consistent
        touched = np.unique(t[(owner[t] == p.part_index).any(axis=1)])
        assert set(touched) == set(g.tolist())                                     # ghosts = exactly the foreign corners
        for nb in p.neighbours:
            q = next(x for x in plans[nb.rank].neighbours if x.rank == p.part_index)
            assert np.array_equal(nb.send_global, q.recv_global)                   # both sides agree on order
            assert np.array_equal(g[nb.recv_start:nb.recv_start + nb.recv_count], nb.recv_global)
            assert nb.contiguous and np.all(nb.send_local < p.n_boundary)          # slabs: plane = contiguous range
        assert len(p.neighbours) <= 2


# ---- a two-layer ghost region: ghosts cross only EVERY OTHER substep -------------------------------------------------
# The algorithm, with the product's depth-2 plan and the oracle as the compute body.  Layers of a rank: owned O, first ghost layer G1
# (shares a tet with O), second layer G2 (shares a tet with G1); tets T1 (touch O) and L2 (touch G1, not O).  Entering an EVEN substep s
# everything local is valid.  Substep s: O and G1 come out right (all their tets are local and had valid inputs), G2 does not.  Substep
# s+1 (ODD): T1 had valid inputs, so O comes out right; L2 did not (no G2), and its per-tet STATE -- quaternion, carried rest shape --
# is now wrong, and so is G1.  Exchange: G1 and G2 take their owners' state after s+1 -- and the L2 tets are EVOLVED AFTER THE FACT:
# from their state after s, with the state of G1 (kept locally) and G2 (sent after s: the early message, which has a whole substep to
# arrive) as they were after s, one more substep.  Messages per two substeps and neighbour: one late (both layers, after the odd
# substep -- the only one on the critical chain) and one early (second layer, after the even substep).
def _worker_deep(rank, world, port, kind, nsteps, out_dir):
    from oracle import OraclePJ
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v, t, owner_fn = _mesh(kind)
    plan = PartitionPlan(t, len(v), world, rank, owner_fn(world), depth=2)
    lv = v[plan.local_to_global_vert]
    quirk = plan.n_local_tets > 0 and plan.local_to_global_tet[0] == 0
    body = OraclePJ(lv, plan.local_tets, PP, slot_quirk=bool(quirk))
    no, ng1 = plan.n_owned, plan.n_ghost1
    G1 = np.arange(no, no + ng1)
    G2 = np.arange(no + ng1, plan.n_local)
    L2 = np.flatnonzero(plan.tet_layer == 1)
    everyone, every_tet = np.arange(plan.n_local), np.arange(plan.n_local_tets)

    def exchange(layers):   # owners' (pos, vel) of the listed ghost layers -> dict ghost local id -> row
        pos, vel = body.pos, body.vel
        reqs, recv = [], []
        for nb in plan.neighbours:
            send = np.concatenate([nb.send_local if 1 in layers else [], nb.send2_local if 2 in layers else []]).astype(np.int64)
            ids = np.concatenate([np.arange(nb.recv_start, nb.recv_start + nb.recv_count) if 1 in layers else [],
                                  np.arange(nb.recv2_start, nb.recv2_start + nb.recv2_count) if 2 in layers else []]).astype(np.int64)
            reqs.append(dist.isend(torch.from_numpy(np.concatenate([pos[send], vel[send]], axis=1).copy()), nb.rank))
            r = torch.empty((len(ids), 6), dtype=torch.float32)
            reqs.append(dist.irecv(r, nb.rank))
            recv.append((ids, r))
        for q in reqs:
            q.wait()
        ids = np.concatenate([i for i, _ in recv]) if recv else np.zeros(0, np.int64)
        rows = np.concatenate([r.numpy() for _, r in recv]) if recv else np.zeros((0, 6), np.float32)
        return ids, rows

    s = 0
    while s < nsteps:
        body.simulate(DT, PP)                                   # EVEN substep: O, G1 and every local tet's state are right afterwards
        s += 1
        g1_pos, g1_vel = body.pos[G1], body.vel[G1]             # G1 after the even substep (local)
        l2_q, l2_el = body.tetState(L2)                         # L2 state after the even substep
        early_ids, early = exchange({2})                        # the EARLY message: second layer after the even substep
        if s == nsteps:                                         # (an odd number of substeps: nothing more to do locally)
            break
        body.simulate(DT, PP)                                   # ODD substep: O right; G1, G2, L2 state wrong
        s += 1
        late_ids, late = exchange({1, 2})                       # the LATE message: both layers after the odd substep
        # evolve the L2 tets after the fact, on a scratch copy of the state: everything back afterwards
        all_pos, all_vel = body.pos, body.vel
        all_q, all_el = body.tetState(every_tet)
        body.writeTets(L2, l2_q, l2_el)
        body.writeParticles(G1, g1_pos, g1_vel)
        body.writeParticles(early_ids, early[:, :3], early[:, 3:])
        body.simulate(DT, PP)
        l2_q, l2_el = body.tetState(L2)                         # = their state after the odd substep
        body.writeParticles(everyone, all_pos, all_vel)
        body.writeTets(every_tet, all_q, all_el)
        body.writeTets(L2, l2_q, l2_el)
        body.writeParticles(late_ids, late[:, :3], late[:, 3:])
    np.save(os.path.join(out_dir, "pos%d.npy" % rank), body.pos[:no])
    np.save(os.path.join(out_dir, "ids%d.npy" % rank), plan.local_to_global_vert[:no])
    # the ghost tets' state must equal the owner's copy of the same tets: save it by global tet id
    q, _ = body.tetState(every_tet)
    np.save(os.path.join(out_dir, "tq%d.npy" % rank), q)
    np.save(os.path.join(out_dir, "tid%d.npy" % rank), plan.local_to_global_tet)
    np.save(os.path.join(out_dir, "tl%d.npy" % rank), plan.tet_layer)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,nsteps", [("slab", 2, 30), ("slab", 3, 30), ("dragon", 2, 24), ("dragon", 3, 12), ("slab", 3, 11)])
def test_two_layer_ghosts_exchanged_every_other_substep_equal_single_process(kind, world, nsteps, tmp_path):
    from oracle import OraclePJ
    mp.spawn(_worker_deep, args=(world, _free_port(), kind, nsteps, str(tmp_path)), nprocs=world, join=True)
    v, t, _ = _mesh(kind)
    mono = OraclePJ(v, t, PP, slot_quirk=True)
    for _ in range(nsteps):
        mono.simulate(DT, PP)
    ref, refq = mono.pos, mono.quats
    seen = np.zeros(len(v), dtype=bool)
    for r in range(world):
        pos, ids = np.load(tmp_path / ("pos%d.npy" % r)), np.load(tmp_path / ("ids%d.npy" % r))
        assert not seen[ids].any()
        seen[ids] = True
        assert np.array_equal(pos.view(np.uint32), ref[ids].view(np.uint32)), (kind, world, r)
        if nsteps % 2 == 0:   # after an exchange every local tet -- second-layer ghost tets included -- carries the state its owner has
            tq, tid = np.load(tmp_path / ("tq%d.npy" % r)), np.load(tmp_path / ("tid%d.npy" % r))
            assert np.array_equal(tq.view(np.uint32), refq[tid].view(np.uint32)), (kind, world, r)
            assert (np.load(tmp_path / ("tl%d.npy" % r)) == 1).any()
    assert seen.all()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_deep_plan_is_symmetric_and_complete(world):
    v, t = make_lattice(4, nz=24)
    owner = slab_owner(4, 24, world)
    plans = [PartitionPlan(t, len(v), world, r, owner, depth=2) for r in range(world)]
    shallow = [PartitionPlan(t, len(v), world, r, owner, depth=1) for r in range(world)]
    for p, q in zip(plans, shallow):
        g = p.local_to_global_vert
        no, ng1 = p.n_owned, p.n_ghost1
        assert p.n_owned == q.n_owned and p.n_owned_tets == q.n_owned_tets
        # the first layer is the depth-1 plan's ghost set; the tets that touch an owned particle are its tets
        assert set(g[no:no + ng1].tolist()) == set(q.local_to_global_vert[q.n_owned:].tolist())
        assert np.array_equal(p.local_to_global_tet[p.tet_layer == 0], q.local_to_global_tet)
        assert np.array_equal(g[p.local_tets], t[p.local_to_global_tet])
        own = owner[t[p.local_to_global_tet]] == p.part_index
        assert np.array_equal(own.any(axis=1), p.tet_layer == 0)
        g1 = set(g[no:no + ng1].tolist())
        for e in np.flatnonzero(p.tet_layer == 1):                                  # second-layer tets: no owned corner, at least one first-layer ghost
            assert any(int(c) in g1 for c in t[p.local_to_global_tet[e]])
        # every tet of a first-layer ghost is local: the partition can update that ghost by itself
        for gv in list(g1)[:50]:
            assert set(np.flatnonzero((t == gv).any(axis=1)).tolist()) <= set(p.local_to_global_tet.tolist())
        for nb in p.neighbours:
            o = next(x for x in plans[nb.rank].neighbours if x.rank == p.part_index)
            assert np.array_equal(nb.send_global, o.recv_global) and np.array_equal(nb.send2_global, o.recv2_global)
            assert np.array_equal(g[nb.recv2_start:nb.recv2_start + nb.recv2_count], nb.recv2_global)
            assert not set(nb.send_global.tolist()) & set(nb.send2_global.tolist())
            assert np.all(nb.send_local < p.n_boundary) and np.all(nb.send2_local < p.n_boundary)
