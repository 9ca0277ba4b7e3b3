"""This rank's body of the benchmark lattice (+ its communicator and, on request, the peer-to-peer halo), and the timed frames."""
import time

import numpy as np

from .common import DT, PP, SUBSTEPS


def slab_owner(nverts, cells, nz, world):
    """vertex -> rank: whole z-planes, ceil(nz / world) cell layers per slab."""
    plane = (cells + 1) * (cells + 1)
    layers = -(-nz // world)
    return np.minimum((np.arange(nverts) // plane) // layers, world - 1).astype(np.int32)


def make_body(args, cells, scaling, rank, world, local_rank, ranks, vote=False, halo=None):
    """This rank's body of the cells^2 x (cells [x world when weak]) lattice (+ communicator when ranks is not None).
    vote=True: creation (local, may fail on one rank alone: memory, ...) is followed by a vote of all ranks BEFORE the collective
    communicator set-up; if any rank failed, every rank returns (None, ..., error text) instead of hanging in the broadcast."""
    from tetsim_amd import SoftBodyHIP, make_lattice
    nz = cells * world if scaling == "weak" else cells
    pp = dict(PP)
    body, verts, tets, err = None, None, None, None
    try:
        verts, tets = make_lattice(cells, nz=nz)
        kw = {}
        if ranks is not None:
            # the stacked lattice is `world` metres long in z: the reference's hard-coded +-2.5 m clamp (SoftbodyGPU.js:347)
            # would squash it, so N > 1 runs honour physicsParams.worldBounds, widened along z
            zext = 0.5 * (nz / cells) + 2.0
            pp["worldBounds"] = [-2.5, -1.0, -zext, 2.5, 10.0, zext]
            kw = dict(part_count=world, part_index=rank, vert_owner=slab_owner(len(verts), cells, nz, world), ref_fixed_bounds=False)
        if args.constant_rest_shape:
            kw["constant_rest_shape"] = True
        if getattr(args, "lean_state", False):
            kw["lean_state"] = True
        if getattr(args, "reference_rotation_exit", False):
            kw["ref_rotation_exit"] = True
        if (halo or args.halo) == "deep" and ranks is not None and world > 1:
            kw["deep_ghosts"] = True
        body = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision=args.precision, device=local_rank, **kw)
    except Exception as e:  # noqa: BLE001
        if not vote:
            raise
        err = "rank %d: %r" % (rank, e)
    if vote and ranks is not None and ranks.min_float(0.0 if err else 1.0) < 1.0:
        if body is not None:
            body.close()
        return None, verts, tets, pp, nz, err or "another rank failed to create its partition"
    if ranks is not None:
        from tetsim_amd import comm_init, comm_unique_id
        uid = ranks.broadcast_bytes(comm_unique_id() if rank == 0 else None, 128)
        comm_init(body, uid, rank, world)
        if (halo or args.halo) in ("p2p", "deep") and world > 1:
            # the peer-to-peer halo on top of the communicator (RCCL keeps carrying the refresh after a dt change): every rank
            # describes its buffers, torch gathers the descriptions, every rank opens its neighbours' (HIP IPC); a local failure is
            # voted on so that no rank steps alone
            from tetsim_amd import p2p_connect, p2p_export
            perr = None
            try:
                blob = p2p_export(body)
            except Exception as e:  # noqa: BLE001
                blob, perr = b"\0" * 512, "rank %d: %r" % (rank, e)
            blobs = ranks.all_gather_bytes(blob, 512)
            if perr is None:
                try:
                    p2p_connect(body, blobs)
                except Exception as e:  # noqa: BLE001
                    perr = "rank %d: %r" % (rank, e)
            if ranks.min_float(0.0 if perr else 1.0) < 1.0:
                if not vote:
                    raise SystemExit("peer-to-peer halo: " + (perr or "another rank could not connect"))
                body.close()
                return None, verts, tets, pp, nz, perr or "another rank could not connect its peer-to-peer halo"
            ranks.barrier()
    return body, verts, tets, pp, nz, None


def timed_frames(body, pp, steps, warmup, ranks):
    """W untimed + K timed frames bracketed by sync + barrier.  Returns (wall seconds of this rank, host seconds this rank spent
    inside the K stepping calls -- the enqueue cost; the calls do not synchronise)."""
    def barrier():
        body.sync()
        if ranks is not None:
            ranks.barrier()

    for _ in range(warmup):
        body.simulateSubsteps(SUBSTEPS, DT, pp)
    barrier()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        body.simulateSubsteps(SUBSTEPS, DT, pp)
        host += time.perf_counter() - h0
    barrier()
    return time.perf_counter() - t0, host
