"""Partitioned bodies (BASELINE config 5's kind) have what unpartitioned ones have: a checkpoint to go back to and something to draw.

* tetsim_save_state / _load_state per partition (missing #2): the reference's whole state is its ping-pong targets
  (SoftbodyGPU.js:49-55); a partition's is its owned particles, its ghosts and its local tets incl. ghost tets.  A restored group
  continues bit for bit -- over the copy transport and over the peer-to-peer halo, saved on an odd substep parity too.
* the embedded visual mesh on a partition (missing #3; SoftbodyGPU.js:424-448, Softbody.js:259-277): every partition skins the
  visual vertices whose tet it owns; the union equals the unpartitioned body's skin bit for bit."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_f32, load_mesh
from tetsim_amd import SoftBodyHIP, TetSimError, group_p2p_connect, group_refresh_final, group_step_n, group_visual_vertex_normals, make_lattice

pytestmark = pytest.mark.gpu
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20


def _slab_owner(nv, cells, parts):
    return np.minimum((np.arange(nv) // (cells + 1) ** 2) * parts // (cells + 1), parts - 1).astype(np.int32)


def _group(v, t, parts, owner, precision, **kw):
    return [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, part_count=parts, part_index=p, vert_owner=owner, **kw)
            for p in range(parts)]


def _gather(bodies, nv, what="pos"):
    out = np.empty((nv, 3), np.float32)
    for b in bodies:
        out[b.ownedIds] = getattr(b, what)
    return out


def _same(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("precision,p2p", [("precise", False), ("fast", False), ("fast", True), ("fast-lean", False), ("fast-lean", True)])
def test_a_restored_group_of_eight_slabs_continues_bit_for_bit(precision, p2p):
    cells, parts = 16, 8
    kw = dict(lean_state=True) if precision == "fast-lean" else {}   # TETSIM_FLAG_LEAN_STATE: the blob holds three corners per tet and the recovered quaternions
    precision = precision.split("-")[0]
    v, t = make_lattice(cells, y0=0.02)          # reaches the floor within the first call: contact is part of the state
    owner = _slab_owner(len(v), cells, parts)
    a = _group(v, t, parts, owner, precision, **kw)
    group_step_n(a, 3, DT, PP)
    if p2p:
        group_p2p_connect(a)
    group_step_n(a, 30, DT, PP)                  # 33 substeps: an ODD parity of the peer-to-peer halo's ghost buffers
    blobs = [b.saveState() for b in a]
    assert len({len(x) for x in blobs}) > 1      # (the end slabs hold one ghost plane, the others two)
    group_step_n(a, 30, DT, PP)
    ref_pos, ref_vel = _gather(a, len(v)), _gather(a, len(v), "vel")
    ref_quat = [b.quats.copy() for b in a]
    # a fresh group -- another process after a lost rank would build exactly this --, transport attached, state loaded
    b = _group(v, t, parts, owner, precision, **kw)
    group_step_n(b, 2, DT, PP)                   # (its own history, and an EVEN parity, before the load: none of it may survive)
    if p2p:
        group_p2p_connect(b)
    for body, blob in zip(b, blobs):
        body.loadState(blob)
    group_step_n(b, 30, DT, PP)
    assert _same(_gather(b, len(v)), ref_pos) and _same(_gather(b, len(v), "vel"), ref_vel)
    for x, q in zip(b, ref_quat):
        assert _same(x.quats, q)
    # the same group once more, in place: back to the checkpoint, same continuation
    for body, blob in zip(a, blobs):
        body.loadState(blob)
    group_step_n(a, 30, DT, PP)
    assert _same(_gather(a, len(v)), ref_pos)
    # a blob belongs to ITS partition of ITS cut
    with pytest.raises(TetSimError, match="another mesh|other options"):
        b[0].loadState(blobs[1])
    other_cut = _group(v, t, 4, _slab_owner(len(v), cells, 4), precision, **kw)
    with pytest.raises(TetSimError, match="another mesh|other options"):
        other_cut[0].loadState(blobs[0])
    mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision, **kw)
    with pytest.raises(TetSimError, match="another mesh|other options"):
        mono.loadState(blobs[0])


def test_two_layer_ghost_regions_have_no_checkpoint():
    v, t = make_lattice(8, y0=0.3)
    deep = _group(v, t, 2, _slab_owner(len(v), 8, 2), "fast", deep_ghosts=True)
    with pytest.raises(TetSimError, match="two-layer"):
        deep[0].saveState()


def _host_skin(pos, tets, vis):
    """The vertex shader's arithmetic (SoftbodyGPU.js:429-435) in numpy f32: ((p0*b0 + p1*b1) + p2*b2) + p3*b3, b3 = 1 - ((b0 + b1) + b2)."""
    e = vis[:, 0].astype(np.int64)
    b0, b1, b2 = (vis[:, k].astype(np.float32)[:, None] for k in (1, 2, 3))
    b3 = np.float32(1.0) - ((b0 + b1) + b2)
    p = [pos[tets[e, k]] for k in range(4)]
    return ((p[0] * b0 + p[1] * b1) + p[2] * b2) + p[3] * b3


@pytest.mark.parametrize("mesh,parts,precision", [("dragon", 3, "precise"), ("dragon", 4, "fast"), ("lattice", 8, "precise"), ("lattice", 8, "fast")])
def test_the_partitions_skins_add_up_to_the_whole_visual_mesh(mesh, parts, precision):
    rng = np.random.default_rng(7)
    if mesh == "dragon":
        v, t = load_mesh("dragon")
        v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
        vis = load_f32("dragon_vis.f32").reshape(-1, 4)      # the reference's own 29,800 rows (Dragon.js:1705)
        owner = None                                          # the library's partitioner: ragged cuts
    else:
        cells = 12
        v, t = make_lattice(cells, y0=0.02)
        owner = _slab_owner(len(v), cells, parts)
        w = rng.dirichlet(np.ones(4), size=20000).astype(np.float32)
        vis = np.concatenate([rng.integers(0, len(t), size=(20000, 1)).astype(np.float32), w[:, :3]], axis=1)
    n0 = rng.normal(size=(len(vis), 3)).astype(np.float32)
    n0 /= np.linalg.norm(n0, axis=1, keepdims=True)
    bodies = _group(v, t, parts, owner, precision)
    for b in bodies:
        b.setVisualMesh(vis, n0)
    kept = np.concatenate([b.visualIds for b in bodies])
    assert len(kept) == len(vis) and np.array_equal(np.sort(kept), np.arange(len(vis)))     # every row in exactly one partition
    assert all(np.all(np.diff(b.visualIds) > 0) for b in bodies if b.numVisVerts > 1)
    for _ in range(3):
        group_step_n(bodies, 10, DT, PP)
    with pytest.raises(TetSimError, match="stale"):
        next(b for b in bodies if b.info.num_neighbours).visualPositions()       # ghost corners: their owners' positions have not been fetched
    group_refresh_final(bodies)
    pos, nrm = np.empty((len(vis), 3), np.float32), np.empty((len(vis), 3), np.float32)
    for b in bodies:
        p, n = b.visualPositions(with_normals=True)
        pos[b.visualIds], nrm[b.visualIds] = p, n
    # against the group's own particle positions, skinned on the host with the shader's f32 arithmetic: bit for bit in either precision
    assert _same(pos, _host_skin(_gather(bodies, len(v)), t, vis))
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1.0).max() < 1e-5
    mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision=precision)
    mono.setVisualMesh(vis, n0)
    mono.simulateSubsteps(30, DT, PP)
    mp, mn = mono.visualPositions(with_normals=True)
    if precision == "precise":      # PRECISE partitions equal the unpartitioned body bit for bit, and so do their skins and rotated normals
        assert _same(pos, mp) and _same(nrm, mn)
    else:
        assert np.abs(pos - mp).max() < 1e-4 and np.abs(nrm - mn).max() < 1e-3
    # computeVertexNormals (Softbody.js:259-277) on the partitioned body: every rank takes the same, global triangle list; a triangle's
    # corners may be skinned by different ranks, so each rank computes ITS rows' normals from the ranks' skins put together -- per vertex
    # the same face normals in the same order as unpartitioned: bit for bit
    if mesh == "dragon":
        tris = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int32).reshape(-1, 3)   # Dragon.js dragonAttachedTriIds
    else:
        tris = rng.integers(0, len(vis), size=(30000, 3)).astype(np.int32)     # (any triangles: corners all over the partitions, degenerate ones included)
    for b in bodies:
        b.setVisualTriangles(tris)
    mono.setVisualTriangles(tris)
    gpos, gnrm = group_visual_vertex_normals(bodies, len(vis))
    assert _same(gpos, pos)
    assert _same(gnrm, mono.visualVertexNormalsFrom(pos))           # the unpartitioned body's kernel on the same positions
    if precision == "precise":
        assert _same(gnrm, mono.visualVertexNormals())              # ... and on its own: the partitions equal it bit for bit
    own = np.empty_like(gnrm)
    for b in bodies:                                                # the rank-by-rank form (what ranks of different processes do)
        own[b.visualIds] = b.visualVertexNormalsFrom(pos)
    assert _same(own, gnrm)
    with pytest.raises(TetSimError, match="put the ranks' skins together"):
        bodies[0].visualVertexNormals()
    # stepping makes the fetched positions stale again
    group_step_n(bodies, 1, DT, PP)
    with pytest.raises(TetSimError, match="stale"):
        next(b for b in bodies if b.info.num_neighbours).visualPositions()
    with pytest.raises(TetSimError, match="stale"):
        group_visual_vertex_normals(bodies, len(vis))


def test_a_group_outlives_a_member_without_touching_it():
    """In-process groups hold plain pointers to their members.  A member that is destroyed waits for its siblings' transfers into its
    ghost ranges and is forgotten by them: a survivor can still save its state; stepping the broken group is refused, not a crash."""
    cells, parts = 8, 3
    v, t = make_lattice(cells, y0=0.3)
    owner = _slab_owner(len(v), cells, parts)
    a = _group(v, t, parts, owner, "fast")
    group_step_n(a, 7, DT, PP)
    a[1].close()                                    # (its neighbours' copies of substep 7 may still be queued: destroy drains them)
    blob = a[0].saveState()
    assert len(blob) > 0 and np.isfinite(a[2].pos).all()
    fresh = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=parts, part_index=1, vert_owner=owner)
    with pytest.raises(TetSimError, match="destroyed or replaced"):
        group_step_n([a[0], fresh, a[2]], 1, DT, PP)
    # a group built anew takes the survivor's checkpoint
    b = _group(v, t, parts, owner, "fast")
    b[0].loadState(blob)
    group_step_n(b, 1, DT, PP)
    assert np.isfinite(_gather(b, len(v))).all()


def test_a_partition_that_keeps_no_row_of_the_visual_mesh():
    cells, parts = 8, 4
    v, t = make_lattice(cells, y0=0.3)
    owner = _slab_owner(len(v), cells, parts)
    bodies = _group(v, t, parts, owner, "fast")
    first_slab = np.flatnonzero((owner[t] == 0).all(axis=1))[:50]       # tets whose four corners lie in slab 0
    vis = np.concatenate([first_slab[:, None].astype(np.float32), np.full((len(first_slab), 3), 0.25, np.float32)], axis=1)
    for b in bodies:
        b.setVisualMesh(vis)
        with pytest.raises(TetSimError, match="already attached"):
            b.setVisualMesh(vis)                                        # (also for the partitions that kept nothing)
    assert [b.numVisVerts for b in bodies] == [len(vis), 0, 0, 0]
    group_step_n(bodies, 5, DT, PP)
    group_refresh_final(bodies)
    assert bodies[3].visualPositions().shape == (0, 3) and len(bodies[3].visualIds) == 0
    assert _same(bodies[0].visualPositions(), _host_skin(_gather(bodies, len(v)), t, vis))
