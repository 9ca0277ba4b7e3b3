#!/usr/bin/env python3
"""The clustered Gauss-Seidel call kernel's chain, wave by wave (1 M-tet lattice): a MUTANT of the library (tools/mutant_lib.py; never
shipped) has every wave of substep 10 of a 20-substep call write three readings of the 100 MHz clock into the volError array -- when it
started (T0), when the last particle it waited for had arrived (T1), when it issued its own hand-over stores (T2).  Per colour: how long
waves sat resident before their data came (T1 - T0), how long their own work took (T2 - T1), and how far the colours' medians are apart
(the pace of the chain: 8 links per substep).

    python tools/nh_link_trace.py build     (no GPU: builds tetsim_amd/libtetsim_hip_nhtrace.so)
    python tools/nh_link_trace.py [cells]   ->  profiles/r06_nh_link_trace.txt
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_nhtrace.so")
EDITS = [
    "nh_kernels.inc", "    if (i >= clusters) return;   // (whole quads leave)\n    const DevParams& P = *d.params;\n    const T dt = P.dt, devC = static_cast<T>(P.d_dev_compliance), volC = static_cast<T>(P.d_vol_compliance);\n    const T a_dev = devC / dt / dt, a_vol = volC / dt / dt, c_off = volC / devC;\n    const uint32_t sub_base = P.epoch + sub * ncol",
    "    if (i >= clusters) return;\n    const long long T0 = wall_clock64();\n    const DevParams& P = *d.params;\n    const T dt = P.dt, devC = static_cast<T>(P.d_dev_compliance), volC = static_cast<T>(P.d_vol_compliance);\n    const T a_dev = devC / dt / dt, a_vol = volC / dt / dt, c_off = volC / devC;\n    const uint32_t sub_base = P.epoch + sub * ncol",
    "nh_kernels.inc", "    float4 keep_a = pa, keep_b = pb;\n    if (fold_a || fold_b) {",
    "    const long long T1 = wall_clock64();\n    float4 keep_a = pa, keep_b = pb;\n    if (fold_a || fold_b) {",
    "nh_kernels.inc", "            if (c == 0u && d.vol_err && last_sub) d.vol_err[ord[j]] = static_cast<double>(ve);\n            if (c < 3u) { sp[a0] = p[0]; sp[a1] = p[1]; sp[a2] = p[2]; sp[a3] = p[3]; }\n        }\n    }\n    if (va >= 0) {\n        float4 r = s_pos[c * 16u + cl];\n        if (((lmask >> c) & 1u) && last_sub) store_wt(d.pos, ua, r);",
    "            if (c == 0u && d.vol_err && last_sub && ve == static_cast<T>(12345.678f)) d.vol_err[ord[j]] = 0.0;\n            if (c < 3u) { sp[a0] = p[0]; sp[a1] = p[1]; sp[a2] = p[2]; sp[a3] = p[3]; }\n        }\n    }\n    {\n        const long long T2 = wall_clock64();\n        if (sub == 10u && lane == 0u && d.vol_err) {\n            unsigned long long* tv = reinterpret_cast<unsigned long long*>(d.vol_err) + (r * 4u + (threadIdx.x >> 6)) * 4ull;\n            tv[0] = static_cast<unsigned long long>(T0) * 16ull + k; tv[1] = static_cast<unsigned long long>(T1); tv[2] = static_cast<unsigned long long>(T2);\n        }\n    }\n    if (va >= 0) {\n        float4 r = s_pos[c * 16u + cl];\n        if (((lmask >> c) & 1u) && last_sub) store_wt(d.pos, ua, r);",
    "nh_kernels.inc", "    if (fold_a) store_wt(d.prev, ua, keep_a);\n    if (fold_b) store_wt(d.prev, ub, keep_b);\n}",
    "    if (fold_a) store_wt(d.prev, ua, keep_a);\n    if (fold_b) store_wt(d.prev, ub, keep_b);\n    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the stores have been acknowledged)\n    if (sub == 10u && lane == 0u && d.vol_err) reinterpret_cast<unsigned long long*>(d.vol_err)[(r * 4u + (threadIdx.x >> 6)) * 4ull + 3ull] = static_cast<unsigned long long>(wall_clock64());\n}",
]
if sys.argv[1:2] == ["build"]:
    sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "tools", "mutant_lib.py"), "nhtrace"] + EDITS + sys.argv[2:]))
os.environ["TETSIM_HIP_LIB"] = os.environ.get("TETSIM_TRACE_LIB", LIB)
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from tetsim_amd import SoftBodyHIP, make_lattice  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 55
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(cells)
dt = (1.0 / 60.0) / 20
b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="fast", order="clustered")
for _ in range(4):
    b.simulateSubsteps(20, dt, pp)
b.sync()
ms = sorted(b.timeSubsteps(20, dt, pp) for _ in range(5))
blob = bytes(b.saveState())
tv = np.frombuffer(blob[-len(t) * 8:], dtype=np.uint64)[:len(t) // 4 * 4].reshape(-1, 4)
tv = tv[tv[:, 3] != 0]
k = (tv[:, 0] & np.uint64(15)).astype(int)
T0 = (tv[:, 0] >> np.uint64(4)).astype(np.int64)
T1, T2, T3 = tv[:, 1].astype(np.int64), tv[:, 2].astype(np.int64), tv[:, 3].astype(np.int64)
base = np.median(T1[k == 0])
us = lambda x: x / 100.0     # 100 MHz ticks
print("lattice %d^3, %d tets, %s: %.1f us per substep; substep 10 of a 20-substep call, %d waves traced (10 ns clock)" % (cells, len(t), os.path.basename(os.environ["TETSIM_HIP_LIB"]), ms[0] * 50, len(tv)))
print("%-6s %6s | %27s | %27s | %24s | %24s | %9s" % ("colour", "waves", "T0 start  p10 / p50 / p90", "T1 arrived p10 / p50 / p90", "resident T1-T0 p10/50/90", "work T2-T1 p10/50/90", "link p50"))
prev = None
for c in range(int(k.max()) + 1):
    m = k == c
    q = lambda a: tuple(us(np.percentile(a[m], p)) for p in (10, 50, 90))
    t0, t1, wait, work = q(T0 - base), q(T1 - base), q(T1 - T0), q(T2 - T1)
    link = "" if prev is None else "%9.2f" % (t1[1] - prev)
    prev = t1[1]
    print("%-6d %6d | %8.2f %8.2f %8.2f  | %8.2f %8.2f %8.2f  | %7.2f %7.2f %7.2f  | %7.2f %7.2f %7.2f  | %s" % ((c, m.sum()) + t0 + t1 + wait + work + (link,)))
# the hand-over itself: a wave's T1 against the LATEST T2 of the waves that produced its particles (colour k -> k + 1 inside the substep)
order = b.tetOrder.astype(np.int64)              # sequential order: colour after colour, cluster after cluster, <= 6 tets each
tets = np.asarray(t)[order]
ncol = int(k.max()) + 1
# clusters: runs of tets in sequential order that share the cube (lattice: 6 tets per cluster, every cluster full)
assert len(t) % 6 == 0
cl_of_tet = np.arange(len(t)) // 6
ncl = len(t) // 6
# colour of a cluster: clusters are laid out colour by colour; the traced waves tell how many waves (16 clusters each) a colour has
waves_per_colour = np.array([(k == c).sum() for c in range(ncol)])
verts = tets.reshape(ncl, 24)
# cluster -> colour: a colour's clusters are consecutive; sizes from the vertex-disjointness (a cluster conflicts with the colour's earlier ones when the colour ends)
col_start = [0]
seen = np.zeros(len(v), dtype=np.int64) - 1
for ci in range(ncl):
    vs = np.unique(verts[ci])
    if (seen[vs] == len(col_start) - 1).any():
        col_start.append(ci)
    seen[vs] = len(col_start) - 1
col_start.append(ncl)
assert len(col_start) == ncol + 1, (len(col_start), ncol)
wave_T1 = {}
wave_T2 = {}
rows = np.flatnonzero(np.frombuffer(blob[-len(t) * 8:], dtype=np.uint64)[:len(t) // 4 * 4].reshape(-1, 4)[:, 3] != 0)   # row = block * 4 + wave in block
first_row = {}
for c in range(ncol):
    first_row[c] = rows[k == c].min()
last_toucher = np.zeros(len(v), dtype=np.int64) - 1      # row of the wave that touched a particle last
gaps = {c: [] for c in range(ncol)}
nprod = {c: [] for c in range(ncol)}
row_index = {r: j for j, r in enumerate(rows)}
for c in range(ncol):
    n = col_start[c + 1] - col_start[c]
    for w0 in range(0, n, 16):
        row = first_row[c] + w0 // 16
        cl = np.arange(col_start[c] + w0, min(col_start[c] + w0 + 16, col_start[c + 1]))
        vs = np.unique(verts[cl])
        if c > 0 and row in row_index:
            prod = np.unique(last_toucher[vs])
            prod = prod[prod >= 0]
            if len(prod) and all(int(q) in row_index for q in prod):
                latest = max(T2[row_index[int(q)]] for q in prod)
                gaps[c].append(T1[row_index[row]] - latest)
                nprod[c].append(len(prod))
        last_toucher[vs] = row
print("the hand-over alone: a wave's T1 minus the LATEST T2 among the waves that produced its particles (store -> visible -> seen), us")
print("%-6s %9s %9s %9s %9s   %s" % ("colour", "p10", "p50", "p90", "max", "producer waves per wave (mean)"))
for c in range(1, ncol):
    g = np.array(gaps[c]) / 100.0
    print("%-6d %9.2f %9.2f %9.2f %9.2f   %.1f" % (c, np.percentile(g, 10), np.percentile(g, 50), np.percentile(g, 90), g.max(), np.mean(nprod[c])))
ack = (T3 - T2) / 100.0
print("a wave's hand-over stores: issued at T2 (+ one LDS read), acknowledged (vmcnt 0) after p10 / p50 / p90 = %.2f / %.2f / %.2f us" % tuple(np.percentile(ack, (10, 50, 90))))
short = (T1 - T0) < 100
print("waves whose data was there within 1 us of their start (they were the late ones, not their data): %.1f %%" % (100.0 * short.mean()))
