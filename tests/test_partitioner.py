"""The built-in vertex partitioner (include/tetsim.h: tetsim_prep_partition; csrc/partitioner.cpp) -- SURVEY.md 8(e) "General meshes:
host-side graph partition".  CPU only.  What a cut must respect is the particle -> incident tets coupling of the reference's Jacobi
average (/root/reference/src/SoftbodyGPU.js:563-577, :306-319): the properties checked here are the ones the halo pays for."""
import numpy as np
import pytest

from conftest import load_mesh
from tetsim_amd import make_lattice
from tetsim_amd._capi import TetSimError
from tetsim_amd.partition import PartitionPlan, index_range_owner, partition, partition_quality


def _renumbered_lattice(cells=16, seed=1):
    v, t = make_lattice(cells)
    perm = np.random.default_rng(seed).permutation(len(v))
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    return np.ascontiguousarray(v[perm]), np.ascontiguousarray(inv[t].astype(np.int32))


def _weights(t, nv):
    return np.bincount(t.reshape(-1), minlength=nv) + 1    # the partitioner's: 1 + valence


MESHES = {"dragon": lambda: load_mesh("dragon"), "lat16_renumbered": _renumbered_lattice, "lat12": lambda: make_lattice(12)}


@pytest.mark.parametrize("mesh", sorted(MESHES))
@pytest.mark.parametrize("parts", [2, 3, 4, 8])
@pytest.mark.parametrize("coords", [False, True])
def test_one_owner_balance_and_fewer_ghosts_than_index_ranges(mesh, parts, coords):
    v, t = MESHES[mesh]()
    nv = len(v)
    own = partition(t, nv, parts, v if coords else None)
    assert own.shape == (nv,) and own.min() >= 0 and own.max() == parts - 1          # one owner per particle, every part used
    assert np.array_equal(own, partition(t, nv, parts, v if coords else None))        # deterministic
    load = np.bincount(own, weights=_weights(t, nv), minlength=parts)
    assert load.max() / load.mean() - 1.0 <= 0.05, load                                # imbalance of what is balanced: <= 5 %
    q, qi = partition_quality(t, nv, parts, own), partition_quality(t, nv, parts, index_range_owner(nv, parts))
    if mesh != "lat12":   # (a lattice in its own z-major order IS slabs when cut into index ranges: nothing to beat for 2 parts)
        assert q["ghost_particle_fraction"] < qi["ghost_particle_fraction"], (q["ghost_particle_fraction"], qi["ghost_particle_fraction"])
        assert q["ghost_tet_fraction"] < qi["ghost_tet_fraction"]
    else:
        assert q["ghost_particle_fraction"] <= 1.02 * qi["ghost_particle_fraction"] + 1e-9
    assert q["local_tet_imbalance"] <= 0.12


def test_quality_table():
    """The figures DESIGN.md 7 quotes (ghost particles / ghost tets, partitioner vs index ranges)."""
    v, t = load_mesh("dragon")
    got = {p: partition_quality(t, len(v), p) for p in (3, 4, 8)}
    rng = {p: partition_quality(t, len(v), p, index_range_owner(len(v), p)) for p in (3, 4, 8)}
    for p in (3, 4, 8):
        assert got[p]["ghost_particle_fraction"] < 0.85 * rng[p]["ghost_particle_fraction"]
    assert got[3]["ghost_particle_fraction"] < 0.06 and got[8]["ghost_particle_fraction"] < 0.26
    rv, rt = _renumbered_lattice()
    for p in (3, 4, 8):
        assert partition_quality(rt, len(rv), p)["ghost_particle_fraction"] < 0.5 * partition_quality(rt, len(rv), p, index_range_owner(len(rv), p))["ghost_particle_fraction"]


@pytest.mark.parametrize("parts", [3, 4])
def test_quality_counts_are_the_plans_counts(parts):
    """tetsim_prep_partition_quality predicts exactly what the partition plans (and therefore tetsim_create) build; and a plan made
    without an owner map uses the partitioner (no coordinates)."""
    v, t = load_mesh("dragon")
    own = partition(t, len(v), parts)
    q = partition_quality(t, len(v), parts, own)
    assert q == partition_quality(t, len(v), parts)       # vert_owner = None: the same map
    for r in range(parts):
        explicit, default = PartitionPlan(t, len(v), parts, r, own), PartitionPlan(t, len(v), parts, r)
        assert np.array_equal(explicit.local_to_global_vert, default.local_to_global_vert)
        assert np.array_equal(explicit.local_to_global_tet, default.local_to_global_tet)
        p = q["parts"][r]
        assert (explicit.n_owned, explicit.n_boundary, explicit.n_local - explicit.n_owned) == (p["owned_particles"], p["boundary_particles"], p["ghost_particles"])
        assert (explicit.n_local_tets, explicit.n_owned_tets, len(explicit.neighbours)) == (p["local_elems"], p["owned_elems"], p["num_neighbours"])
    assert sum(p["owned_elems"] for p in q["parts"]) == len(t) and sum(p["owned_particles"] for p in q["parts"]) == len(v)


def test_edge_cases():
    v, t = make_lattice(3)
    assert not partition(t, len(v), 1).any()                                             # one part: everything is part 0
    cloud = partition(np.zeros((0, 4), np.int32), 10, 3)                                  # a tet-less particle cloud still splits evenly
    assert sorted(np.bincount(cloud, minlength=3).tolist()) == [3, 3, 4]
    two = np.concatenate([t, t + len(v)])                                                 # two disconnected copies: each part stays inside one
    own = partition(two, 2 * len(v), 2)
    assert len(set(own[:len(v)])) == 1 and len(set(own[len(v):])) == 1 and own[0] != own[-1]
    many = partition(t, len(v), len(v) + 5)                                               # more parts than particles: still one owner each, in range
    assert many.min() >= 0 and many.max() < len(v) + 5
    assert partition(np.zeros((0, 4), np.int32), 0, 4).shape == (0,)
    with pytest.raises(TetSimError):
        partition(np.int32([[0, 1, 2, 99]]), 4, 2)                                        # id out of range
    with pytest.raises(TetSimError):
        partition(t, len(v), 0)
    bad = v.copy(); bad[3, 1] = np.nan
    with pytest.raises(TetSimError):
        partition(t, len(v), 2, bad)
