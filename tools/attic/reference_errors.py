"""Prints max |dx| (m) of the device polar solver against the reference-GLSL golden vectors at every recorded substep."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, load_f32, load_mesh
from test_oracle_golden_glsl import replay
from tetsim_amd import SoftBodyHIP
g = json.load(open(os.path.join(GOLDEN, "golden_gpu.json")))
cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "cases_gpu.json")))}
for name, c in cases.items():
    for label, kw in (("precise", dict(precision="precise")), ("fast-blocked", dict(precision="fast")), ("fast-gather", dict(precision="fast", gather=True)),
                      ("fast-const-rest", dict(precision="fast", constant_rest_shape=True))):
        v, t = load_mesh(c["mesh"])
        body = SoftBodyHIP(v, t, None, dict(c["params"]), solver="polar", ref_grab_texel=True, **kw)
        errs = {}
        replay(body, c, g["cases"][name], lambda s: errs.__setitem__(s, float(np.abs(body.pos - load_f32(f"{name}_gpu_pos_{s}.f32").reshape(-1, 3)).max())),
               lambda gid, p: body.setGrab(gid, p), body.endGrab)
        print("%-12s %-16s %s" % (name, label, "  ".join("%d: %.1e" % kv for kv in errs.items())), flush=True)
