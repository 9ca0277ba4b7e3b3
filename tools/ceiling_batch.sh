#!/usr/bin/env bash
# The inputs of the bench line's rocprof / traffic / ceiling objects and of profiles/r06z_tet_kernel_ceiling.txt, on a GPU box:
#   bash tools/ceiling_batch.sh <tag>   ->  gpurun_out/<tag>/{stats/, pmc/, iteration_floor.txt, tet_kernel_ceiling.{txt,json}, pmc_traffic.json, bench.json}
set -u
TAG=${1:-ceiling}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-beyond-mall > "$OUT/stats.log" 2>&1 )
python tools/kernel_launch_stats.py "$OUT/stats/s_kernel_trace.csv" > "$OUT/kernel_windows.txt" 2>&1
# counters of the EQUAL-WORK kernel: the headline body with the reference's rotation threshold (nine iterations in every tet)
BENCH_ARGS="--reference-rotation-exit --no-replay" timeout 900 bash tools/pmc_run.sh $TAG/pmc > "$OUT/pmc.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_counters.txt" 2>&1
python tools/pmc_traffic.py "$OUT/pmc" "$OUT/pmc_traffic.json" > /dev/null 2>&1
timeout 900 bash tools/iteration_floor.sh > "$OUT/iteration_floor.txt" 2>&1
python tools/tet_kernel_ceiling.py --pmc "$OUT/pmc" --iters "$OUT/iteration_floor.txt" --write "$OUT/tet_kernel_ceiling.json" > "$OUT/tet_kernel_ceiling.txt" 2>&1
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json; cp "$OUT/tet_kernel_ceiling.json" profiles/tet_kernel_ceiling.json
python - "$OUT" <<'PY'
import json, sys, os
sys.path.insert(0, ".")
from tetsim_amd import library_info
json.dump({"csv": "bench_kernel_stats.csv", "kernel_sha": library_info()["kernel_sha"]}, open(os.path.join(sys.argv[1], "bench_kernel_stats.json"), "w"))
PY
cp "$OUT/stats/s_kernel_stats.csv" profiles/bench_kernel_stats.csv; cp "$OUT/bench_kernel_stats.json" profiles/bench_kernel_stats.json
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 600 python -m pytest tests/test_bench_gpu.py tests/test_gpu_frame_kernel.py -q -x -m gpu > "$OUT/pytest_bench_frame.log" 2>&1
gzip -9f "$OUT/stats/s_kernel_trace.csv"; rm -f "$OUT"/pmc/pass*/*kernel_trace.csv
cat "$OUT/iteration_floor.txt" "$OUT/tet_kernel_ceiling.txt" "$OUT/kernel_windows.txt"; tail -5 "$OUT/pytest_bench_frame.log"; tail -3 "$OUT/bench.err"
