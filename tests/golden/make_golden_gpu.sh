#!/usr/bin/env bash
# Golden vectors for the POLAR (WebGL) solver: runs the REFERENCE's own SoftbodyGPU.js +
# MultiTargetGPUComputationRenderer.js + vendored three.js under Node, with GL executed by Mesa softpipe (swrast_dri.so, GALLIUM_DRIVER=softpipe)
# (oracle/glsl_ref).  BUILD container only: needs /root/reference, node and Mesa's swrast_dri.so.
# Reference files are copied to a scratch directory (never into the repo); only DATA is written here.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF=${REF:-/root/reference}
"$ROOT/oracle/glsl_ref/build.sh" > /dev/null
SCRATCH=$(mktemp -d /tmp/tetsim_refgpu.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
mkdir -p "$SCRATCH/src" "$SCRATCH/node_modules/three/build"
cp "$REF/src/SoftbodyGPU.js" "$REF/src/MultiTargetGPUComputationRenderer.js" "$SCRATCH/src/"
cp "$REF/node_modules/three/build/three.module.js" "$SCRATCH/node_modules/three/build/"
echo '{"type":"module"}' > "$SCRATCH/package.json"
echo '{"type":"module"}' > "$SCRATCH/node_modules/three/package.json"
# softpipe: plain fp32 GLSL interpreter.  llvmpipe lowers the shaders' default-precision (lowp) sampler results to
# fp16, which GLSL ES permits but no desktop WebGL stack does.
GALLIUM_DRIVER=${GALLIUM_DRIVER:-softpipe} node "$HERE/make_golden_gpu.mjs" "$SCRATCH" "$HERE" "$ROOT/oracle/_ref/mesa_gl.node" "$@"
