"""The JavaScript boundary against the REAL three.js (missing #3): `.edgeMesh` / `.visMesh` of SoftBodyHIP built with
the three r160 module the reference itself vendors (/root/reference/node_modules/three), not with a stand-in -- LineSegments / Mesh
instances, layer 1 against a layer-1 Raycaster, userData, the position attribute aliasing the caller's array, version counters behind
`needsUpdate`, and a Raycaster.intersectObjects -> startGrab round trip as Grabber.start does (Softbody.js:36-57, 440-456; main.js:60-68).
Build container only (the reference is absent on the GPU box), CPU only: the native addon is replaced by a 2-tet stand-in inside the node
script.  three.js is imported from a scratch copy; nothing of it is committed or shipped."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

THREE_MODULE = "/root/reference/node_modules/three/build/three.module.js"
NODE = shutil.which("node")


@pytest.mark.skipif(NODE is None or not os.path.exists(THREE_MODULE), reason="needs node and the reference's vendored three.js (build container only)")
def test_display_objects_against_the_reference_s_own_three_js(tmp_path):
    build = tmp_path / "node_modules" / "three" / "build"
    build.mkdir(parents=True)
    shutil.copy(THREE_MODULE, build / "three.module.js")
    (tmp_path / "node_modules" / "three" / "package.json").write_text('{"type":"module"}')
    r = subprocess.run([NODE, os.path.join(ROOT, "tetsim_amd", "node", "test_three_boundary.mjs"), str(build / "three.module.js")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "THREE_BOUNDARY_OK r160" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
