#!/usr/bin/env bash
# Step 2 of golden generation (run in the BUILD container only: needs /root/reference and node).
# Copies the reference's CPU solver + data + three.js into a scratch directory (never into the repo),
# marks both package scopes as ES modules (SURVEY.md §8(c) recipe) and runs make_golden.mjs, which
# imports the reference's SoftBody and writes DATA fixtures (inputs + expected outputs) next to itself.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=${REF:-/root/reference}
SCRATCH=$(mktemp -d /tmp/tetsim_ref.XXXXXX)
trap 'rm -rf "$SCRATCH"' EXIT
mkdir -p "$SCRATCH/src" "$SCRATCH/node_modules/three/build"
cp "$REF/src/Softbody.js" "$REF/src/Dragon.js" "$SCRATCH/src/"
cp "$REF/node_modules/three/build/three.module.js" "$SCRATCH/node_modules/three/build/"
echo '{"type":"module"}' > "$SCRATCH/package.json"
echo '{"type":"module"}' > "$SCRATCH/node_modules/three/package.json"
node "$HERE/make_golden.mjs" "$SCRATCH" "$HERE"
