#!/usr/bin/env bash
# Builds the headless Mesa GL binding (swrast_dri.so; the goldens are recorded with its softpipe driver) used ONLY to generate the polar-solver
# golden vectors (tests/golden/make_golden_gpu.sh), and the table of GL enumerants webgl2_context.mjs needs, taken from the image's own GL
# headers.  Output goes to oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
mkdir -p "$OUT"
g++ -std=c++17 -O1 -shared -fPIC -I/usr/include/node "$HERE/mesa_gl.cc" -o "$OUT/mesa_gl.node" -ldl
python3 - "$OUT/gl_constants.json" <<'PY'
import json, re, sys
out = {}
for h in ("/usr/include/GL/gl.h", "/usr/include/GL/glext.h"):
    for m in re.finditer(r"^#define\s+GL_([A-Za-z0-9_]+)\s+(0x[0-9A-Fa-f]+|\d+)\s*$", open(h).read(), re.M):
        out.setdefault(m.group(1), int(m.group(2), 0))
json.dump(out, open(sys.argv[1], "w"))
PY
echo "$OUT/mesa_gl.node"
