#!/usr/bin/env bash
# Round-2 measurement batch (run ON the GPU box through gpurun):  bash tools/r02_perf.sh <tag>
set -u
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
{
python -m tetsim_amd.build --ablation
python -m tetsim_amd.build --variant planes -DTETSIM_VAR_POS_PLANES
python -m tetsim_amd.build --variant t128 -DTETSIM_TILE=128
python -m tetsim_amd.build --variant t128p -DTETSIM_TILE=128 -DTETSIM_VAR_POS_PLANES
} > "$OUT/build.log" 2>&1
echo "== A/B of kernel variants (bench.py, 40 frames)" > "$OUT/ab.txt"
timeout 600 python tools/ab_lib.py tetsim_amd/libtetsim_hip.so tetsim_amd/libtetsim_hip_planes.so tetsim_amd/libtetsim_hip_t128.so tetsim_amd/libtetsim_hip_t128p.so >> "$OUT/ab.txt" 2>&1
echo "== per-tile phase timeline (ablation build)" > "$OUT/trace.txt"
timeout 200 python tools/trace_tet.py >> "$OUT/trace.txt" 2>&1
echo "== iteration ablation" > "$OUT/iters.txt"
timeout 300 python tools/ab_iters.py >> "$OUT/iters.txt" 2>&1
for g in 0 1; do
  echo "## TETSIM_HALO_GRAPH=$g" >> "$OUT/loopback.txt"
  TETSIM_HALO_GRAPH=$g timeout 300 python tools/loopback_rank.py >> "$OUT/loopback.txt" 2>&1
done
timeout 900 python -m pytest tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_edge_cases.py tests/test_bench_gpu.py tests/test_node_boundary.py -m gpu -q 2>&1 | tail -30 > "$OUT/pytest.log"
TETSIM_RECORD_ERRORS=$OUT/spin.jsonl timeout 300 python -m pytest tests/test_gpu_polar.py -m gpu -q -k spinning > "$OUT/spin.log" 2>&1
timeout 900 bash tools/mutation_check.sh "$OUT/mutation" > "$OUT/mutation.txt" 2>&1
tail -5 "$OUT/ab.txt"; tail -8 "$OUT/loopback.txt"; tail -5 "$OUT/pytest.log"; cat "$OUT/spin.jsonl"; tail -3 "$OUT/mutation.txt"
