"""The Node.js side of the drop-in boundary: N-API shim (tetsim_amd/node/tetsim_napi.cc) + SoftBodyHIP.js."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

NODE = shutil.which("node")
NODE_DIR = os.path.join(ROOT, "tetsim_amd", "node")


def _addon():
    from tetsim_amd.node.build_addon import build_addon
    try:
        return build_addon()
    except RuntimeError as e:
        pytest.skip(str(e))


@pytest.mark.skipif(NODE is None, reason="node is not installed on this host")
def test_addon_loads_and_fails_loudly_without_gpu():
    _addon()
    js = ("const a=require(%r); const keys=Object.keys(a).sort().join(',');"
          "console.log('KEYS '+keys); console.log('ABI '+a.load(%r));"
          "const own=a.partition(null,5,new Int32Array([0,1,2,3,1,2,3,4]),2);console.log('OWNERS '+Array.from(own).join(''));"   # host-only: works without a GPU
          "console.log('QUALITY '+a.partitionQuality(new Int32Array([0,1,2,3,1,2,3,4]),5,2,own).map(p=>p.ownedParticles+'/'+p.ghostParticles).join(' '));"
          "try{a.create(new Float32Array([0,0,0,1,0,0,0,1,0,0,0,1]),new Int32Array([0,1,2,3]),{});console.log('CREATED');}"
          "catch(e){console.log('THROWN '+e.message);}") % (os.path.join(NODE_DIR, "tetsim_napi.node"),
                                                         os.path.join(ROOT, "tetsim_amd", "libtetsim_hip.so"))
    out = subprocess.run([NODE, "-e", js], capture_output=True, text=True, timeout=120).stdout
    assert ("KEYS batchLayout,commInit,commUniqueId,create,createBatch,createFromFile,destroy,groupRefreshFinal,groupStepN,haloRefreshFinal,info,libraryInfo,load,loadState,mapPositions,mapQuats,ownedIds,partition,partitionQuality,readMesh,"
            "readPositions,readQuats,readVelocities,readVisualMesh,readVisualVertexNormals,readVolError,refreshPositions,refreshQuats,saveState,"
            "setGrab,setVisualMesh,setVisualTriangles,startGrab,step,stepN,sync,visualIds,visualVertexNormalsFrom") in out
    assert "ABI 5" in out
    owners = out.split("OWNERS ")[1].split()[0]
    assert len(owners) == 5 and set(owners) == {"0", "1"} and "QUALITY " in out   # the partitioner through N-API: one owner per particle, both parts used
    assert "CREATED" in out or "no CPU fallback" in out  # on a GPU host creation succeeds; otherwise it must throw


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node is not installed on this host")
def test_softbodyhip_js_bit_exact_vs_reference_goldens():
    """SoftBodyHIP.js (reference constructor + simulate/endFrame/grab surface) over N-API on the GPU:
    Neo-Hookean PRECISE == Softbody.js goldens bit for bit; polar frame loop + grab."""
    _addon()
    import tempfile
    from conftest import load_f32, load_mesh
    from tetsim_amd.meshfile import write_mesh
    with tempfile.TemporaryDirectory() as tmp:   # a .tetsim container for SoftBodyHIP.fromFile
        v, t = load_mesh("dragon")
        mesh = os.path.join(tmp, "dragon.tetsim")
        write_mesh(mesh, v, t, vis_verts=load_f32("dragon_vis.f32"))
        r = subprocess.run([NODE, os.path.join(NODE_DIR, "test_softbody.js")], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, TETSIM_TEST_MESH=mesh))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "node boundary ok" in r.stdout and "fromFile" in r.stdout
