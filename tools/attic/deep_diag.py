"""Development: where a two-layer-ghost run (TETSIM_FLAG_DEEP_GHOSTS) leaves the monolithic body.  Steps two substeps at a time and
prints, per call, the worst position error of owned particles against the monolithic body and the worst difference between a ghost
tet's quaternion and its owner's copy, by tet layer.   python tools/attic/deep_diag.py dragon 3 [grab|nograb] [gid]      DEEP=0: the one-layer peer-to-peer halo instead (the yardstick)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
os.environ.setdefault("TETSIM_HALO_TIMEOUT_MS", "3000")
from conftest import load_mesh
from tetsim_amd import SoftBodyHIP, group_step_n, group_p2p_connect, make_lattice

PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (1.0 / 60.0) / 20
kind, n = sys.argv[1], int(sys.argv[2])
grab = len(sys.argv) > 3 and sys.argv[3] == "grab"
if kind == "dragon":
    v, t = load_mesh("dragon")
    v = v - np.float32([0.0, v[:, 1].min() - 0.01, 0.0])
    owner = None
else:
    cells = 16
    v, t = make_lattice(cells, y0=0.02)
    owner = np.minimum((np.arange(len(v)) // (cells + 1) ** 2) * n // (cells + 1), n - 1).astype(np.int32)
parts = [SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", part_count=n, part_index=p, vert_owner=owner, deep_ghosts=os.environ.get("DEEP", "1") == "1") for p in range(n)]
if os.environ.get("DEEP", "1") != "1":
    group_step_n(parts, 1, DT, PP)   # (a group exists after its first step)
group_p2p_connect(parts)
mono = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast")
if os.environ.get("DEEP", "1") != "1":
    mono.simulateSubsteps(1, DT, PP)
gid = int(sys.argv[4]) if len(sys.argv) > 4 else int(parts[0].ownedIds[0])
print("grab particle", gid, "owner part", [i for i, b in enumerate(parts) if gid in set(b.ownedIds.tolist())])
for b in parts:
    print("part owned", len(b.ownedIds), "local tets", b.info.local_elems)
for call in range(24):
    if grab and call == 4:
        for b in parts + [mono]:
            b.setGrab(gid, [float(v[gid, 0]) + 0.15, float(v[gid, 1]) + 0.3, float(v[gid, 2])])
    if grab and call == 16:
        for b in parts + [mono]:
            b.endGrab()
    group_step_n(parts, 2, DT, PP)
    mono.simulateSubsteps(2, DT, PP)
    pos = np.zeros_like(mono.pos)
    for b in parts:
        pos[b.ownedIds] = b.pos
    err = np.abs(pos - mono.pos).max(axis=1)
    mq = mono.quats
    worst = []
    for b in parts:
        q, m = b.quats, mq[b.localTets]
        d = np.minimum(np.abs(q - m).max(axis=1), np.abs(q + m).max(axis=1))   # (q and -q are one rotation)
        worst.append(float(d.max()))
    print("call %2d  pos err %.3g at particle %d   tet quats vs monolithic per part: %s" % (call, err.max(), int(err.argmax()), " ".join("%.3g" % w for w in worst)))
