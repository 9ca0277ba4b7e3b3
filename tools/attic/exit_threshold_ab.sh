#!/usr/bin/env bash
# Exit threshold of the FAST rotation iteration: which calibrated errors move, and what the kernel gains.
#   bash tools/attic/exit_threshold_ab.sh <outdir> <libA.so> <libB.so> ...      (libraries under tetsim_amd/, built with
#   python -m tetsim_amd.build --variant NAME -DTETSIM_ROT_EXIT_W2=<squared threshold>; run ON the GPU box)
# 1. every FAST parity check in record mode against each library -> <outdir>/errors_<lib>.jsonl, one table of observed errors
#    beside the committed calibration (tests/golden/tolerances.json); 2. in-run A/B of the bench line (tools/ab_lib.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$1; shift
mkdir -p "$OUT"
cd "$ROOT"
TESTS="tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_full_size.py tests/test_gpu_random_meshes.py tests/test_gpu_frame_kernel.py tests/test_gpu_p2p_halo.py tests/test_gpu_edge_cases.py"
for lib in "$@"; do
  rm -f "$OUT/errors_$lib.jsonl"
  TETSIM_RECORD_ERRORS="$OUT/errors_$lib.jsonl" TETSIM_HIP_LIB="$ROOT/tetsim_amd/$lib" timeout 1500 python -m pytest $TESTS -m gpu -q -p no:cacheprovider 2>&1 | tail -3 > "$OUT/pytest_$lib.log"
done
python - "$OUT" "$ROOT/tests/golden/tolerances.json" "$@" > "$OUT/errors_table.txt" <<'PY'
import json, os, sys
out, tabp, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
tab = json.load(open(tabp))["checks"]
obs = {}
for lib in libs:
    d = {}
    try:
        for line in open(os.path.join(out, "errors_%s.jsonl" % lib)):
            r = json.loads(line); d[r["label"]] = max(d.get(r["label"], 0.0), r["observed"])
    except FileNotFoundError:
        pass
    obs[lib] = d
labels = [k for k in obs[libs[0]] if any(abs(obs[l].get(k, 0.0) - obs[libs[0]][k]) > 0 for l in libs[1:])]
print("checks whose observed error differs between the libraries (%d of %d); 'calibrated' / 'allowed' = tests/golden/tolerances.json" % (len(labels), len(obs[libs[0]])))
print("%-86s %10s %10s %10s " % ("check", "stated", "calibrated", "allowed") + " ".join("%14s" % l.replace("libtetsim_hip", "").replace(".so", "")[-14:] for l in libs))
summary = {l: [0, 0, 0.0] for l in libs}
for k in labels:
    c = tab.get(k, {"observed": float("nan"), "allowed": float("nan"), "stated": float("nan")})
    cells = []
    for l in libs:
        o = obs[l].get(k, float("nan"))
        mark = ""
        if c["allowed"] == c["allowed"] and o > c["allowed"]:
            mark = "F"; summary[l][0] += 1
        if o > c.get("stated", float("inf")):
            mark = "S"; summary[l][1] += 1
        if c["observed"] and c["observed"] == c["observed"]:
            summary[l][2] = max(summary[l][2], o / c["observed"])
        cells.append("%13.3g%1s" % (o, mark))
    print("%-86s %10.3g %10.3g %10.3g " % (k[:86], c.get("stated", float("nan")), c["observed"], c["allowed"]) + " ".join(cells))
print()
for l in libs:
    print("%-40s above the calibrated allowance (F): %3d   above the STATED bound (S): %3d   largest observed / calibrated: %.2f" % (l, summary[l][0], summary[l][1], summary[l][2]))
PY
tail -8 "$OUT/errors_table.txt"
python tools/ab_lib.py $(for l in "$@"; do echo tetsim_amd/$l; done) -- --no-beyond-mall > "$OUT/ab_lib.txt" 2>&1
cat "$OUT/ab_lib.txt"
