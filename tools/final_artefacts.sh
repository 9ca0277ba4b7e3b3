set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01i; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err); echo "rocprof rc=$?"
timeout 900 bash tools/pmc_run.sh r01i/pmc > $O/pmc.log 2>&1; echo "pmc rc=$?"
python tools/pmc_summary.py gpurun_out/r01i/pmc > $O/pmc_counters.txt 2>&1
timeout 300 python tools/ab_iters.py > $O/ab_iters.txt 2>&1
for c in 80 110; do timeout 300 python bench.py --no-cpu-baseline --steps 40 --cells $c 2>/dev/null | tail -1 > $O/bench_cells$c.json; done
timeout 300 python bench.py --no-cpu-baseline --constant-rest-shape 2>/dev/null | tail -1 > $O/bench_constant_rest_shape.json
tail -1 $O/bench.json | cut -c1-200; cat $O/ab_iters.txt; head -4 $O/prof/b_kernel_stats.csv | cut -c1-160
