#!/usr/bin/env bash
# A round's evidence batch (run ON the GPU box):  bash tools/evidence_batch.sh <tag>
# 1. the suite against the FROZEN tolerance table, then a calibration run (record mode) whose report must not be refused
#    (tools/tolerance_report.py: no label above its ceiling without a reason) -- the table itself is updated in the build container,
# 2. both mutants against the table, 3. tools/ceiling_batch.sh: rocprofv3 stats of the bench command, PMC passes on the equal-work
#    kernel, the product kernel rebuilt with 0 / 3 / 6 iterations (tools/mutant_lib.py beforehand), tet_kernel_ceiling, the bench line,
# 4. bench over 200 frames, 5. Neo-Hookean counters and timings, Dragon, size sweeps, 6. the loopback rank's halo slack.
set -u
TAG=${1:-batch}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -25 > "$OUT/pytest_with_table.log"
rm -f "$OUT/errors.jsonl"
TETSIM_RECORD_ERRORS=$OUT/errors.jsonl timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > "$OUT/pytest_calibration.log"
python tools/tolerance_report.py "$OUT/errors.jsonl" > "$OUT/tolerances.txt" 2>&1; echo "tolerance_report rc=$?" >> "$OUT/tolerances.txt"
timeout 900 bash tools/mutation_check.sh "$OUT/mutation" > "$OUT/mutation.txt" 2>&1
MUTATION=iters timeout 900 bash tools/mutation_check.sh "$OUT/mutation_iters" > "$OUT/mutation_iters.txt" 2>&1
timeout 1500 bash tools/ceiling_batch.sh $TAG > "$OUT/ceiling_batch.log" 2>&1
timeout 300 python bench.py > "$OUT/bench_200.json" 2>> "$OUT/bench.err"
BENCH_ARGS="--solver neohookean" timeout 900 bash tools/pmc_run.sh $TAG/pmc_nh > "$OUT/pmc_nh.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc_nh" > "$OUT/pmc_counters_nh.txt" 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_nh" -o s -- python "$ROOT/bench.py" --solver neohookean --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/stats_nh.log" 2>&1 )
rm -f "$OUT"/stats_nh/*kernel_trace.csv "$OUT"/pmc_nh/pass*/*kernel_trace.csv
timeout 300 python tools/dragon_time.py > "$OUT/dragon.txt" 2>&1
timeout 300 python tools/nh_time.py 55 clustered > "$OUT/nh_time.txt" 2>&1
timeout 400 python tools/size_sweep.py > "$OUT/size_sweep.txt" 2>&1
timeout 400 python tools/nh_size_sweep.py > "$OUT/nh_size_sweep.txt" 2>&1
timeout 900 bash tools/halo_slack.sh "$OUT/halo_slack.txt" > /dev/null 2>&1
timeout 200 python tools/stream_peak.py > "$OUT/stream_peak.txt" 2>&1
tail -3 "$OUT/pytest_with_table.log"; tail -2 "$OUT/tolerances.txt"; tail -6 "$OUT/mutation.txt"; tail -4 "$OUT/mutation_iters.txt"; cat "$OUT/iteration_floor.txt"; head -c 900 "$OUT/bench.json"
