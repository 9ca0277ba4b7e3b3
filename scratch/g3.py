import sys, numpy as np
sys.path.insert(0,'/root/repo')
from tetsim_amd import SoftBodyHIP, group_step_n, make_lattice
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT=(1/60)/20
for cells,n in ((6,3),(8,4),(20,3)):
    v,t=make_lattice(cells,nz=cells*n,y0=0.02)
    plane=(cells+1)**2
    owner=np.minimum((np.arange(len(v))//plane)//cells,n-1).astype(np.int32)
    kw=dict(solver='polar',precision='fast',ref_fixed_bounds=False)
    mono=SoftBodyHIP(v,t,None,dict(PP),**kw)
    parts=[SoftBodyHIP(v,t,None,dict(PP),part_count=n,part_index=i,vert_owner=owner,**kw) for i in range(n)]
    for c in range(6):
        mono.simulateSubsteps(7,DT,PP); group_step_n(parts,7,DT,PP)
    got=np.full_like(mono.pos,np.nan)
    for p in parts: got[p.ownedIds]=p.pos
    print(cells,n,'group max|dx|',np.abs(got-mono.pos).max(), [ (p.info.local_elems) for p in parts])
