#!/usr/bin/env python3
"""Where the 42 us of a clustered Gauss-Seidel substep go (1 M-tet lattice, nh_call_kernel): the product library timed next to two
MUTANTS of it that compute garbage on purpose (tools/mutant_lib.py; never shipped) --
  nhnowait:  no look at a stamp ever waits (every cluster solves at once on whatever is there): the kernel's THROUGHPUT floor;
  nhnosolve: the hand-overs as they are, the six tet solves of a cluster replaced by nothing: the CHAIN of hand-overs alone.

    python tools/mutant_lib.py nhnowait nh_kernels.inc '<pend line>' 'bool pend_a = false, pend_b = false;' ... ; python tools/nh_chain_ablation.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r)
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = make_lattice(int(sys.argv[1]))
dt = (1.0 / 60.0) / 20
b = SoftBodyHIP(v, t, None, dict(pp), solver="neohookean", precision="fast", order="clustered")
b.simulateSubsteps(20, dt, pp); b.sync()
ms = sorted(b.timeSubsteps(20, dt, pp) for _ in range(9))
print("%%-10s frame(20) best %%.3f ms median %%.3f ms = %%.1f us/substep" %% (sys.argv[2], ms[0], ms[4], ms[0] * 50), flush=True)
''' % ROOT
cells = sys.argv[1] if len(sys.argv) > 1 else "55"
for name in ["product"] + sys.argv[2:] + ["product"]:
    env = dict(os.environ, TETSIM_HALO_TIMEOUT_MS="2000")
    if name != "product":
        env["TETSIM_HIP_LIB"] = os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_%s.so" % name)
    subprocess.run([sys.executable, "-c", CHILD, cells, name], env=env, timeout=600)
