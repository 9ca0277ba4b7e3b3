#!/usr/bin/env bash
# A round's evidence batch (run ON the GPU box):  bash tools/evidence_batch.sh <tag>      (CALIBRATE=0 keeps tests/golden/tolerances.json as it is)
# 1. tolerance calibration of the whole GPU suite (TETSIM_RECORD_ERRORS) -> tolerances.json, 2. the suite again WITH the table,
# 3. mutation check against the table, 4. PMC passes (polar + NH) -> pmc_traffic.json, 5. bench lines, 6. rocprofv3 kernel stats of the same command,
# 7. the loopback rank with both halo transports and 0 / 10 / 20 us of injected delay, 8. the frame kernel's phase stamps, Neo-Hookean timings.
set -u
TAG=${1:-batch}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
python -m tetsim_amd.build > "$OUT/build.log" 2>&1
python -m tetsim_amd.build --ablation >> "$OUT/build.log" 2>&1
rm -f "$OUT/errors.jsonl"
TETSIM_RECORD_ERRORS=$OUT/errors.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > "$OUT/pytest_calibration.log"
python tools/tolerance_report.py "$OUT/errors.jsonl" --write "$OUT/tolerances.json" > "$OUT/tolerances.txt" 2>&1
[ "${CALIBRATE:-1}" = 1 ] && cp "$OUT/tolerances.json" tests/golden/tolerances.json
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > "$OUT/pytest_with_table.log"
timeout 900 bash tools/mutation_check.sh "$OUT/mutation" > "$OUT/mutation.txt" 2>&1
MUTATION=iters timeout 900 bash tools/mutation_check.sh "$OUT/mutation_iters" > "$OUT/mutation_iters.txt" 2>&1
timeout 900 bash tools/pmc_run.sh $TAG/pmc > "$OUT/pmc.log" 2>&1
BENCH_ARGS="--solver neohookean" timeout 900 bash tools/pmc_run.sh $TAG/pmc_nh > "$OUT/pmc_nh.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc" > "$OUT/pmc_counters.txt" 2>&1
python tools/pmc_summary.py "$OUT/pmc_nh" > "$OUT/pmc_counters_nh.txt" 2>&1
python tools/pmc_traffic.py "$OUT/pmc" "$OUT/pmc_traffic.json" > /dev/null 2>&1
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json   # keyed by kernel_sha: the bench lines below attach it to roofline.traffic
timeout 400 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
timeout 300 python bench.py > "$OUT/bench_200.json" 2>> "$OUT/bench.err"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-beyond-mall > "$OUT/stats.log" 2>&1 )
python tools/trace_windows.py "$OUT/stats/s_kernel_trace.csv" > "$OUT/kernel_windows.txt" 2>&1; grep -o '"value": [0-9.]*, "unit": "M tet-solves/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": [0-9.]*' "$OUT/stats.log" | head -1 | sed 's/^/  the line of this profiled run: /' >> "$OUT/kernel_windows.txt"
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_nh" -o s -- python "$ROOT/bench.py" --solver neohookean --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stats_nh.log" 2>&1 )
# 7. one interior rank in loopback (the stand-in for a multi-GPU rank): RCCL transfer and peer-to-peer stores, 0 / 10 / 20 us of injected delay
timeout 900 bash tools/halo_slack.sh "$OUT/halo_slack.txt" > /dev/null 2>&1
# 7b. round 4: how many rotation iterations move anything, the four-lane kernels of small bodies, the streaming probes
timeout 900 python tools/rotation_iterations.py > "$OUT/rotation_iterations.txt" 2> "$OUT/rotation_iterations.err"
timeout 600 bash tools/quad_lanes_report.sh > "$OUT/quad_lanes.txt" 2>&1
timeout 200 python tools/stream_peak.py > "$OUT/stream_peak.txt" 2>&1
# 8. small bodies and the Gauss-Seidel solver
timeout 120 python tools/frame_trace.py 20 > "$OUT/frame_trace.txt" 2>&1
timeout 300 python tools/dragon_time.py > "$OUT/dragon.txt" 2>&1
timeout 300 python tools/nh_time.py 55 clustered > "$OUT/nh_time.txt" 2>&1
timeout 400 python tools/size_sweep.py > "$OUT/size_sweep.txt" 2>&1
timeout 400 python tools/nh_size_sweep.py > "$OUT/nh_size_sweep.txt" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head
tail -3 "$OUT/pytest_calibration.log"; tail -6 "$OUT/pytest_with_table.log"; tail -12 "$OUT/mutation.txt"; head -c 600 "$OUT/bench.json"
