import sys, time, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conftest import load_f32, load_mesh, GOLDEN
from tetsim_amd import SoftBodyHIP
PP = dict(gravity=-9.81, friction=1000.0, density=1000.0, devCompliance=1e-5, volCompliance=0.0, worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
v, t = load_mesh("dragon"); vis = load_f32("dragon_vis.f32").reshape(-1, 4)
tris = np.fromfile(os.path.join(GOLDEN, "dragon_vistris.u16"), dtype="<u2").astype(np.int32).reshape(-1, 3)
b = SoftBodyHIP(v, t, None, dict(PP), vis, solver="neohookean", precision="precise"); b.setVisualTriangles(tris)
b.simulateSubsteps(10, 1/600, PP); b.visualVertexNormals(); b.visualPositions()
t0 = time.perf_counter()
for _ in range(200): b.visualPositions()
t1 = time.perf_counter()
for _ in range(200): b.visualVertexNormals()
t2 = time.perf_counter()
print("skinned positions read-back %.3f ms, vertex normals (skin + normals + read-back) %.3f ms per call" % ((t1 - t0) / 200 * 1e3, (t2 - t1) / 200 * 1e3))
