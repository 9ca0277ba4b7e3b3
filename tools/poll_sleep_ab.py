import os, sys, subprocess, json
ROOT="/root/repo"
code='''
import sys,os,time
sys.path.insert(0,"%s")
from tetsim_amd import SoftBodyHIP, make_lattice
pp = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v,t = make_lattice(55); dt=(1/60)/20
def run(**kw):
    b = SoftBodyHIP(v,t,None,dict(pp),**kw)
    for _ in range(5): b.simulateSubsteps(20,dt,pp)
    b.sync()
    fall = sorted(b.timeSubsteps(20,dt,pp) for _ in range(15))[7]
    for _ in range(25): b.simulateSubsteps(20,dt,pp)
    fl = sorted(b.timeSubsteps(20,dt,pp) for _ in range(9))[4]
    b.close(); return fall*50, fl*50
print("polar ref   fall %%.2f floor %%.2f us/substep" %% run(solver="polar",precision="fast"))
print("polar lean  fall %%.2f floor %%.2f" %% run(solver="polar",precision="fast",lean_state=True))
print("nh clustered fast %%.2f %%.2f" %% run(solver="neohookean",precision="fast",order="clustered"))
''' % ROOT
for rep in range(2):
    for lib in ("libtetsim_hip.so","libtetsim_hip_s2.so","libtetsim_hip_s32.so"):
        env=dict(os.environ, TETSIM_HIP_LIB=os.path.join(ROOT,"tetsim_amd",lib))
        r=subprocess.run([sys.executable,"-c",code],env=env,capture_output=True,text=True)
        print(lib); print(r.stdout.strip() or r.stderr[-500:], flush=True)
