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
from tetsim_amd.partition import PartitionPlan, slab_owner

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def _mesh(kind):
    if kind == "slab":
        v, t = make_lattice(4, nz=9, y0=0.03)
        return v, t, lambda world: slab_owner(4, 9, world)
    v, t = load_mesh("dragon")
    return v, t, lambda world: None  # index-range ownership: ragged, non-contiguous halos


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
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("kind,world", [("slab", 2), ("slab", 3), ("dragon", 2)])
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
