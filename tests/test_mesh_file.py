"""The .tetsim mesh container: format round trip and validation on CPU, create-from-file parity on GPU."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, load_f32, load_mesh
from tetsim_amd import SoftBodyHIP, TetSimError, make_lattice
from tetsim_amd.meshfile import MeshFile, greedy_colours, write_mesh

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])


def _dragon(tmp_path, **kw):
    v, t = load_mesh("dragon")
    vis = load_f32("dragon_vis.f32")
    path = os.path.join(tmp_path, "dragon.tetsim")
    write_mesh(path, v, t, vis_verts=vis, **kw)
    return path, v, t, vis


def test_round_trip_all_sections(tmp_path):
    v, t = load_mesh("lat4")
    edges = np.arange(40, dtype=np.int32) % len(v)
    vis = np.array([[0, .25, .25, .25], [3, .1, .2, .3]], dtype=np.float32)
    tris = np.array([[0, 1, 1]], dtype=np.int32)
    colour, n = greedy_colours(t, len(v))
    owner = (np.arange(len(v)) * 3 // len(v)).astype(np.int32)
    path = os.path.join(tmp_path, "lat4.tetsim")
    write_mesh(path, v, t, edge_ids=edges, vis_verts=vis, vis_tri_ids=tris, tet_colour=colour, vert_owner=owner, part_count=3)
    assert os.path.getsize(path) % 64 == 0
    with MeshFile(path) as m:
        assert (m.num_particles, m.num_elems, m.part_count) == (len(v), len(t), 3)
        assert np.array_equal(m.verts.view(np.uint32), v.view(np.uint32)) and np.array_equal(m.tets, t)
        assert np.array_equal(m.edge_ids.ravel(), edges) and np.array_equal(m.vis_verts, vis) and np.array_equal(m.vis_tri_ids, tris)
        assert np.array_equal(m.tet_colour, colour) and np.array_equal(m.vert_owner, owner)
    # minimal file: optional sections are absent, an empty tet list is legal
    write_mesh(path, v, np.zeros((0, 4), np.int32))
    with MeshFile(path) as m:
        assert m.num_elems == 0 and m.tets.shape == (0, 4) and m.edge_ids is None and m.tet_colour is None and m.part_count == 0


def test_stored_colouring_is_proper(tmp_path):
    v, t = load_mesh("dragon")
    colour, n = greedy_colours(t, len(v))
    assert colour.min() == 0 and colour.max() == n - 1
    for c in range(n):   # tets of one colour are vertex-disjoint
        ids = t[colour == c].ravel()
        assert len(np.unique(ids)) == len(ids)


def test_rejects_damaged_files(tmp_path):
    v, t = load_mesh("lat4")
    path = os.path.join(tmp_path, "a.tetsim")
    write_mesh(path, v, t)
    raw = bytearray(open(path, "rb").read())

    def expect(mutated, what):
        bad = os.path.join(tmp_path, "bad.tetsim")
        open(bad, "wb").write(bytes(mutated))
        with pytest.raises(TetSimError, match=what):
            MeshFile(bad)

    expect(b"not a mesh", "too short")
    m = bytearray(raw); m[0] = ord("X"); expect(m, "bad magic")
    m = bytearray(raw); struct.pack_into("<I", m, 8, 99); expect(m, "unsupported .tetsim version")
    expect(raw[:-64], "truncated")
    m = bytearray(raw); struct.pack_into("<Q", m, 64 + 24, len(raw)); expect(m, "out of bounds")        # first section's offset
    m = bytearray(raw); struct.pack_into("<I", m, 64 + 12, 5); expect(m, "wrong type or shape")           # verts with 5 columns
    # a tet that references a particle past the end
    with MeshFile(path) as mf:
        pass
    sec_tets_off = struct.unpack_from("<Q", raw, 64 + 32 + 24)[0]
    m = bytearray(raw); struct.pack_into("<i", m, sec_tets_off, len(v)); expect(m, "out of range")
    with pytest.raises(TetSimError, match="cannot open"):
        MeshFile(os.path.join(tmp_path, "missing.tetsim"))
    with pytest.raises(TetSimError, match="vert_owner entry out of"):
        write_mesh(path, v, t, vert_owner=np.full(len(v), 7, np.int32), part_count=2)


def test_cli_pack_and_info(tmp_path, capsys):
    from conftest import GOLDEN
    from tetsim_amd.meshfile import _main
    out = os.path.join(tmp_path, "d.tetsim")
    _main(["pack", "-o", out, "--verts", os.path.join(GOLDEN, "dragon_verts.f32"), "--tets", os.path.join(GOLDEN, "dragon_tets.i32"),
           "--vis", os.path.join(GOLDEN, "dragon_vis.f32"), "--colour", "--parts", "4"])
    _main(["info", out])
    text = capsys.readouterr().out
    assert "particles 1234  tets 3840  part_count 4" in text and "tet_colour   (3840,)" in text


@pytest.mark.gpu
def test_create_from_file_equals_create_from_arrays(tmp_path):
    path, v, t, vis = _dragon(tmp_path)
    dt = (1 / 60) / 10
    for kw in (dict(solver="neohookean", precision="precise"), dict(solver="polar", precision="precise"), dict(solver="polar", precision="fast")):
        a = SoftBodyHIP(v, t, None, dict(PP), visVerts=vis, **kw)
        b = SoftBodyHIP.fromFile(path, dict(PP), **kw)
        assert (b.numParticles, b.numElems, b.numVisVerts) == (a.numParticles, a.numElems, a.numVisVerts)
        a.simulateSubsteps(30, dt, PP)
        b.simulateSubsteps(30, dt, PP)
        assert np.array_equal(a.pos.view(np.uint32), b.pos.view(np.uint32)), kw
        assert np.array_equal(a.visualPositions().view(np.uint32), b.visualPositions().view(np.uint32)), kw
    # a container that also carries the visual triangle list gives the body its vertex normals (computeVertexNormals on the device)
    tris = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int32).reshape(-1, 3)
    path2, _, _, _ = _dragon(os.path.join(tmp_path, "."), vis_tri_ids=tris)
    a = SoftBodyHIP(v, t, None, dict(PP), vis, tris, solver="neohookean", precision="precise")
    b = SoftBodyHIP.fromFile(path2, dict(PP), solver="neohookean", precision="precise")
    for body in (a, b):
        body.simulateSubsteps(10, dt, PP)
    assert np.array_equal(a.visualVertexNormals().view(np.uint32), b.visualVertexNormals().view(np.uint32))
    assert np.array_equal(b.visualVertexNormals().view(np.uint32), load_f32("dragon_visnormal_10.f32").reshape(-1, 3).view(np.uint32))


@pytest.mark.gpu
def test_stored_colouring_and_partition_are_used(tmp_path):
    v, t = load_mesh("dragon")
    dt = (1 / 60) / 10
    # (1) a caller-supplied colouring: any labelling is safe, and it equals a sequential solve in "sorted by colour" order
    rng = np.random.RandomState(5)
    silly = rng.randint(0, 7, len(t)).astype(np.int32)           # NOT a proper colouring
    path = os.path.join(tmp_path, "silly.tetsim")
    write_mesh(path, v, t, tet_colour=silly)
    body = SoftBodyHIP.fromFile(path, dict(PP), solver="neohookean", precision="precise", order="coloured")
    order = body.tetOrder
    assert np.array_equal(order, np.argsort(silly, kind="stable"))
    from oracle import OracleNH
    orc = OracleNH(v, t[order], PP)                                # the reference algorithm fed the permuted sequence
    for _ in range(20):
        body.simulate(dt, PP)
        orc.simulate(dt, PP)
    assert np.array_equal(body.pos.view(np.uint32), orc.pos.view(np.uint32))
    # the built-in colouring stored in a file gives the same body as order="coloured" without a file
    greedy, _ = greedy_colours(t, len(v))
    write_mesh(path, v, t, tet_colour=greedy)
    a = SoftBodyHIP.fromFile(path, dict(PP), solver="neohookean", order="coloured")
    b = SoftBodyHIP(v, t, None, dict(PP), solver="neohookean", order="coloured")
    assert np.array_equal(a.tetOrder, b.tetOrder) and a.info.num_levels == b.info.num_levels
    # (2) a stored partition map
    lv, lt = make_lattice(6)
    owner = ((lv[:, 2] - lv[:, 2].min()) / (np.ptp(lv[:, 2]) + 1e-6) * 2).astype(np.int32).clip(0, 1)
    write_mesh(path, lv, lt, vert_owner=owner, part_count=2)
    parts = [SoftBodyHIP.fromFile(path, dict(PP), solver="polar", part_count=2, part_index=i) for i in range(2)]
    ref = [SoftBodyHIP(lv, lt, None, dict(PP), solver="polar", part_count=2, part_index=i, vert_owner=owner) for i in range(2)]
    for p, r in zip(parts, ref):
        assert np.array_equal(p.ownedIds, r.ownedIds)
    with pytest.raises(TetSimError, match="stored partition map is for 2 parts"):
        SoftBodyHIP.fromFile(path, dict(PP), solver="polar", part_count=3, part_index=0)
