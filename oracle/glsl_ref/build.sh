#!/usr/bin/env bash
# Builds the headless Mesa (swrast_dri.so; the goldens are recorded with its softpipe driver) GL binding used ONLY to generate the polar-solver golden vectors
# (tests/golden/make_golden_gpu.sh).  Output goes to oracle/_ref/ (git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../_ref"
mkdir -p "$OUT"
g++ -std=c++17 -O1 -shared -fPIC -I/usr/include/node "$HERE/mesa_gl.cc" -o "$OUT/mesa_gl.node" -ldl
echo "$OUT/mesa_gl.node"
