"""Synthetic Kuhn-6 cube lattices and mesh fixture IO.

The reference ships one mesh (``src/Dragon.js``); BASELINE.json's roofline configs use a
synthetic lattice instead.  The generator is deterministic (no RNG):

* ``n`` cells per side (``nz`` may differ for slab-stacked multi-GPU lattices), spacing ``h = 1/n``;
* vertex ``(i, j, k)`` at ``((i - n/2) h, y0 + j h, (k - n/2) h)``, index ``i + (n+1) (j + (n+1) k)``;
* every cell is split into the 6 Kuhn tetrahedra (the 6 monotone paths (0,0,0)->(1,1,1));
  vertices 2 and 3 are swapped where the signed volume would be negative, so every rest
  volume is ``+h^3/6``;
* tets are stored cell-major (cell ``i`` fastest), 6 consecutive tets per cell.
"""
import itertools
import numpy as np

_PERMS = list(itertools.permutations(range(3)))  # the 6 axis orders


def make_lattice(n, nz=None, y0=0.5, dtype=np.float32):
    """Return ``(verts[Nv,3] float32, tets[Nt,4] int32)`` for an ``n x n x nz`` cell lattice."""
    nz = n if nz is None else nz
    h = 1.0 / n
    nx1, ny1, nz1 = n + 1, n + 1, nz + 1
    k, j, i = np.meshgrid(np.arange(nz1), np.arange(ny1), np.arange(nx1), indexing="ij")
    verts = np.stack([(i - n / 2.0) * h, y0 + j * h, (k - nz / 2.0) * h], axis=-1)
    verts = verts.reshape(-1, 3).astype(dtype)

    ck, cj, ci = np.meshgrid(np.arange(nz), np.arange(n), np.arange(n), indexing="ij")
    base = np.stack([ci.ravel(), cj.ravel(), ck.ravel()], axis=-1)  # [Nc,3] cell origin

    def vid(p):
        return p[:, 0] + nx1 * (p[:, 1] + ny1 * p[:, 2])

    tets = np.empty((base.shape[0], 6, 4), dtype=np.int64)
    for t, perm in enumerate(_PERMS):
        p = base.copy()
        corner = [vid(p)]
        for ax in perm:
            p = p.copy()
            p[:, ax] += 1
            corner.append(vid(p))
        # signed volume of the path tet = sign of the permutation (same for every cell)
        e = np.zeros((3, 3))
        for c, ax in enumerate(perm):
            e[c:, ax] = 1.0  # edge vectors p1-p0, p2-p0, p3-p0 in unit-cell coordinates
        if np.linalg.det(e) < 0:
            corner[2], corner[3] = corner[3], corner[2]
        tets[:, t, :] = np.stack(corner, axis=-1)
    return verts, tets.reshape(-1, 4).astype(np.int32)


def save_mesh(prefix, verts, tets):
    np.ascontiguousarray(verts, dtype="<f4").tofile(prefix + "_verts.f32")
    np.ascontiguousarray(tets, dtype="<i4").tofile(prefix + "_tets.i32")


def load_mesh(prefix):
    verts = np.fromfile(prefix + "_verts.f32", dtype="<f4").reshape(-1, 3)
    tets = np.fromfile(prefix + "_tets.i32", dtype="<i4").reshape(-1, 4)
    return verts, tets
