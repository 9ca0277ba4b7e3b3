#!/usr/bin/env bash
# Copy what tools/evidence_batch.sh <tag> left under gpurun_out/<tag>/ (scratch, merged back from the GPU box) into profiles/ (tracked):
#   bash tools/collect_evidence.sh <tag>
# The three files the bench line attaches by kernel_sha (bench_kernel_stats.{csv,json}, pmc_traffic.json, tet_kernel_ceiling.json) keep
# their fixed names; everything else is named <tag>_*.
set -eu
TAG=${1:?tag}
cd "$(dirname "$0")/.."
S=gpurun_out/$TAG; P=profiles
cpf() { if [ -s "$S/$1" ]; then cp "$S/$1" "$P/$2"; else echo "missing: $S/$1" >&2; fi; }
cpf bench.json ${TAG}_bench.json
cpf bench_200.json ${TAG}_bench_200steps.json
cpf stats/s_kernel_stats.csv ${TAG}_bench_kernel_stats.csv
cpf stats/s_kernel_stats.csv bench_kernel_stats.csv
cpf bench_kernel_stats.json bench_kernel_stats.json
cpf pmc_traffic.json pmc_traffic.json
cpf tet_kernel_ceiling.json tet_kernel_ceiling.json
cpf kernel_windows.txt ${TAG}_bench_kernel_windows.txt
cpf dragon.txt ${TAG}_dragon.txt
cpf halo_slack.txt ${TAG}_halo_slack.txt
cpf iteration_floor.txt ${TAG}_iteration_floor.txt
cpf mutation.txt ${TAG}_mutation.txt
cpf mutation_iters.txt ${TAG}_mutation_iters.txt
cpf stats_nh/s_kernel_stats.csv ${TAG}_neohookean_kernel_stats.csv
cpf nh_time.txt ${TAG}_neohookean_time.txt
cpf nh_size_sweep.txt ${TAG}_nh_size_sweep.txt
cpf size_sweep.txt ${TAG}_size_sweep.txt
cpf stream_peak.txt ${TAG}_stream_peak.txt
cpf pmc_counters_nh.txt ${TAG}_pmc_counters_neohookean.txt
cpf pmc_counters.txt ${TAG}_pmc_counters_reference_threshold.txt
cpf pytest_with_table.log ${TAG}_pytest_with_table.log
cpf tet_kernel_ceiling.txt ${TAG}_tet_kernel_ceiling.txt
cpf tolerances.txt ${TAG}_tolerances.txt
python - <<'PY'
import json
from tetsim_amd import build
_, ker = build.source_shas()
for f in ("bench_kernel_stats.json", "pmc_traffic.json", "tet_kernel_ceiling.json"):
    k = json.load(open("profiles/" + f)).get("kernel_sha")
    print("%-28s kernel_sha %s %s" % (f, k, "(this tree)" if k == ker else "STALE: this tree is " + ker))
PY
