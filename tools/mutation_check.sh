#!/usr/bin/env bash
# Mutation check of the FAST tolerances (run ON the GPU box):  bash tools/mutation_check.sh [outdir]
# Builds a MUTATED copy of the library (see the mutant list below) and runs the GPU parity tests against it through
# TETSIM_HIP_LIB, first in record mode (mutant_vs_product.txt: which calibrated errors moved), then for real.
# Expected for `bias`: tests FAIL -- the calibrated tolerances see a systematic two-ulp error.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/mutation}
mkdir -p "$OUT"
MUT=$(mktemp -d /tmp/tetsim_mut.XXXXXX)
mkdir -p "$MUT/tetsim_amd" && cp -r "$ROOT/tetsim_amd/csrc" "$MUT/tetsim_amd/csrc" && cp "$ROOT/tetsim_amd/build.py" "$ROOT/tetsim_amd/__init__.py" "$MUT/tetsim_amd/" && cp -r "$ROOT/include" "$MUT/include"
rm -rf "$MUT/tetsim_amd/csrc/obj" "$MUT/tetsim_amd/csrc/obj_ablation"
# mutants (MUTATION=...):
#   bias   (default) a +2 ulp systematic bias on the quaternion normalisation factor of the FAST path (pj_math.inc normalize4)
#   cos    the reference's cos(h) = sin(h + 1.57) quirk "fixed" to sin(h + pi/2) (SoftbodyGPU.js:106-110) in the FAST iteration
#   iters  8 instead of 9 rotation iterations in the blocked kernel and in the four-lane kernels of small bodies
case "${MUTATION:-bias}" in
  bias)  FILE=pj_math.inc;    FROM='const float r = __builtin_amdgcn_rsqf(d);'; TO='const float r = __builtin_amdgcn_rsqf(d) * 1.00000024f; /* MUTATION */' ;;
  cos)   FILE=pj_math.inc;    FROM='__builtin_amdgcn_sinf(rev + 0.24987326f)';  TO='__builtin_amdgcn_sinf(rev + 0.25f) /* MUTATION */' ;;
  iters) FILE=pj_lab.h;       FROM='#define TETSIM_ROTATION_ITERATIONS 9';     TO='#define TETSIM_ROTATION_ITERATIONS 8 /* MUTATION */'
         FILE2=pj_quad.hip;   FROM2='for (int iter = 1; iter < 9; iter++)';      TO2='for (int iter = 1; iter < 8; iter++) /* MUTATION */' ;;
  *) echo "unknown MUTATION"; exit 2 ;;
esac
grep -nF "$FROM" "$MUT/tetsim_amd/csrc/$FILE" > "$OUT/mutated_line.txt" || { echo "mutation target not found"; exit 2; }
mutate() { python - "$1" "$2" "$3" <<'PY'
import sys
p, a, b = sys.argv[1:4]
s = open(p).read()
assert a in s
open(p, "w").write(s.replace(a, b))
PY
}
mutate "$MUT/tetsim_amd/csrc/$FILE" "$FROM" "$TO"
if [ -n "${FILE2:-}" ]; then   # (the four-lane kernels of small bodies carry the same constant)
  grep -nF "$FROM2" "$MUT/tetsim_amd/csrc/$FILE2" >> "$OUT/mutated_line.txt" || { echo "second mutation target not found"; exit 2; }
  mutate "$MUT/tetsim_amd/csrc/$FILE2" "$FROM2" "$TO2"
fi
(cd "$MUT" && python -c "import sys; sys.path.insert(0, '.'); import importlib.util as u; s = u.spec_from_file_location('b', 'tetsim_amd/build.py'); m = u.module_from_spec(s); s.loader.exec_module(m); print(m.build(force=True))") > "$OUT/build.log" 2>&1 || { echo "mutant build failed"; tail -5 "$OUT/build.log"; exit 2; }
cd "$ROOT"
rm -f "$OUT/mutant_errors.jsonl"
TETSIM_RECORD_ERRORS="$OUT/mutant_errors.jsonl" TETSIM_HIP_LIB="$MUT/tetsim_amd/libtetsim_hip.so" python -m pytest tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_full_size.py tests/test_gpu_random_meshes.py -m gpu -q -k "fast or spinning or constant or lattice_1m or lattice_8m or random or rotation" -p no:cacheprovider > /dev/null 2>&1
python - "$OUT/mutant_errors.jsonl" "$ROOT/tests/golden/tolerances.json" > "$OUT/mutant_vs_product.txt" <<'PY'
import json, sys
mut = {}
for line in open(sys.argv[1]):
    r = json.loads(line); mut[r["label"]] = max(mut.get(r["label"], 0.0), r["observed"])
tab = json.load(open(sys.argv[2]))["checks"]
print("%-84s %10s %10s %10s" % ("check", "product", "mutant", "allowed"))
for k, m in mut.items():
    c = tab.get(k)
    if c and (m > c["observed"] * 1.5 or (c["allowed"] and m > c["allowed"])):
        print("%-84s %10.3g %10.3g %10.3g%s" % (k[:84], c["observed"], m, c["allowed"], "  <-- FAILS" if c["allowed"] and m > c["allowed"] else ""))
PY
TETSIM_HIP_LIB="$MUT/tetsim_amd/libtetsim_hip.so" python -m pytest tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_full_size.py tests/test_gpu_random_meshes.py -m gpu -q -k "fast or spinning or constant or lattice_1m or lattice_8m or random or rotation" -p no:cacheprovider 2>&1 | tail -40 > "$OUT/pytest_mutant.log"
grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_mutant.log" | tail -30
if grep -q "failed" "$OUT/pytest_mutant.log"; then echo "MUTATION DETECTED (${MUTATION:-bias}): the tests above fail"; else echo "MUTATION SURVIVED (${MUTATION:-bias})"; fi
