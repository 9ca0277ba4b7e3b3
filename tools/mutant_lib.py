#!/usr/bin/env python3
"""A library built from a MUTATED copy of the sources (no GPU needed):  python tools/mutant_lib.py NAME FILE 'FROM' 'TO' [FILE 'FROM' 'TO' ...]
-> tetsim_amd/libtetsim_hip_NAME.so (git-ignored; travels to the GPU box with the snapshot; load it through TETSIM_HIP_LIB).
The product sources are not touched: the copy lives in a temporary directory.  Used for the iteration ablation of the PRODUCT kernel
(tools/iteration_floor.sh: pj_lab.h: `#define TETSIM_ROTATION_ITERATIONS 9` -> 0 / 3 / 6) -- the development build's run-time knob compiles to a slower
kernel and cannot give the product's memory floor."""
import importlib.util
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, edits = sys.argv[1], sys.argv[2:]
assert edits and len(edits) % 3 == 0, __doc__
with tempfile.TemporaryDirectory(prefix="tetsim_mut.") as mut:
    os.makedirs(os.path.join(mut, "tetsim_amd"))
    shutil.copytree(os.path.join(ROOT, "tetsim_amd", "csrc"), os.path.join(mut, "tetsim_amd", "csrc"), ignore=shutil.ignore_patterns("obj*"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(mut, "include"))
    for f in ("build.py", "__init__.py"):
        shutil.copy(os.path.join(ROOT, "tetsim_amd", f), os.path.join(mut, "tetsim_amd", f))
    for i in range(0, len(edits), 3):
        p = os.path.join(mut, "tetsim_amd", "csrc", edits[i])
        s = open(p).read()
        assert edits[i + 1] in s, "mutation target not found in %s: %s" % (edits[i], edits[i + 1])
        open(p, "w").write(s.replace(edits[i + 1], edits[i + 2]))
    spec = importlib.util.spec_from_file_location("mutbuild", os.path.join(mut, "tetsim_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = m.build(force=True)
    out = os.path.join(ROOT, "tetsim_amd", "libtetsim_hip_%s.so" % name)
    shutil.copy(lib, out)
    print(out)
