#!/usr/bin/env bash
# Mutation check of the FAST tolerances (run ON the GPU box):  bash tools/mutation_check.sh [outdir]
# Builds a copy of the library in which the quaternion normalisation of the FAST path lacks its Newton step (a ~1-ulp
# systematic bias in v_rsq_f32's result, pj_math.inc normalize4) and runs the GPU parity tests against it through
# TETSIM_HIP_LIB.  Expected: at least one test FAILS -- the tolerances are tight enough to see a one-ulp bias.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/mutation}
mkdir -p "$OUT"
MUT=$(mktemp -d /tmp/tetsim_mut.XXXXXX)
mkdir -p "$MUT/tetsim_amd" && cp -r "$ROOT/tetsim_amd/csrc" "$MUT/tetsim_amd/csrc" && cp "$ROOT/tetsim_amd/build.py" "$ROOT/tetsim_amd/__init__.py" "$MUT/tetsim_amd/" && cp -r "$ROOT/include" "$MUT/include"
rm -rf "$MUT/tetsim_amd/csrc/obj" "$MUT/tetsim_amd/csrc/obj_ablation"
grep -n "r = r \* (1.5f - 0.5f \* d \* r \* r);" "$MUT/tetsim_amd/csrc/pj_math.inc" > "$OUT/mutated_line.txt" || { echo "mutation target not found"; exit 2; }
sed -i 's|        r = r \* (1.5f - 0.5f \* d \* r \* r);|        /* MUTATION: Newton step removed */|' "$MUT/tetsim_amd/csrc/pj_math.inc"
(cd "$MUT" && python -c "import sys; sys.path.insert(0, '.'); import importlib.util as u; s = u.spec_from_file_location('b', 'tetsim_amd/build.py'); m = u.module_from_spec(s); s.loader.exec_module(m); print(m.build(force=True))") > "$OUT/build.log" 2>&1 || { echo "mutant build failed"; tail -5 "$OUT/build.log"; exit 2; }
cd "$ROOT"
TETSIM_HIP_LIB="$MUT/tetsim_amd/libtetsim_hip.so" python -m pytest tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_full_size.py tests/test_gpu_random_meshes.py -m gpu -q -k "fast or spinning or constant or lattice_1m or lattice_8m or random" -p no:cacheprovider 2>&1 | tail -40 > "$OUT/pytest_mutant.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_mutant.log" | tail -30
if grep -q "failed" "$OUT/pytest_mutant.log"; then echo "MUTATION DETECTED: the one-ulp normalisation bias fails the tests above"; else echo "MUTATION SURVIVED: tolerances too loose"; fi
