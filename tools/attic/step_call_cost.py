"""What a host that keeps the reference's loop pays (main.js:79-84: one simulate() per substep): host time per tetsim_step call and substep time
until the queue is drained, against tetsim_step_n, on the Dragon with both solvers.  python tools/attic/step_call_cost.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ctypes as C
from conftest import load_mesh
from tetsim_amd import SoftBodyHIP
from tetsim_amd import _capi as capi
from tetsim_amd.softbody import make_params
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v,t=load_mesh("dragon")
for solver, kw, n in (("polar", dict(precision="fast"), 20), ("neohookean", dict(precision="precise", order="coloured"), 10)):
    b=SoftBodyHIP(v,t,None,dict(PP),solver=solver,**kw); dt=(1/60)/n
    L=b._L; h=b._h; par=make_params(PP); pref=C.byref(par)
    for _ in range(200): L.tetsim_step(h, C.c_double(dt), pref)
    b.sync()
    N=4000
    t0=time.perf_counter()
    for _ in range(N): L.tetsim_step(h, C.c_double(dt), pref)
    t1=time.perf_counter(); b.sync(); t2=time.perf_counter()
    print("%-11s tetsim_step x %d: host %.2f us per call (ctypes included), until drained %.2f us per substep" % (solver, N, (t1-t0)/N*1e6, (t2-t0)/N*1e6))
    t0=time.perf_counter()
    for _ in range(N//n): L.tetsim_step_n(h, n, C.c_double(dt), pref)
    t1=time.perf_counter(); b.sync(); t2=time.perf_counter()
    print("%-11s tetsim_step_n(%d):     host %.2f us per substep, until drained %.2f us per substep" % (solver, n, (t1-t0)/N*1e6, (t2-t0)/N*1e6))
