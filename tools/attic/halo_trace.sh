#!/bin/bash
# kernel timeline of the loopback rank (eager and graph replay), with and without an added wire delay:  bash tools/attic/halo_trace.sh [outdir]
# LOOPBACK_P2P=1 traces the peer-to-peer halo instead of the RCCL transfer
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"; O=${1:-$PWD/gpurun_out/halo_trace}; mkdir -p $O; : > $O/timeline.txt
export TMPDIR=/tmp
for g in 0 1; do for d in 0 20; do
  rm -rf /tmp/tr; 
  LOOPBACK_SKIP_MONO=1 LOOPBACK_CALLS=5 LOOPBACK_REPS=1 TETSIM_HALO_GRAPH=$g TETSIM_DEBUG_LOOPBACK_DELAY_US=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python tools/loopback_rank.py > $O/trace_g${g}_d${d}.log 2>&1
  f=$(find /tmp/tr -name '*kernel_trace.csv' | head -1)
  echo "== graph=$g delay=$d ($f)" >> $O/timeline.txt
  python tools/attic/halo_timeline.py "$f" 3 >> $O/timeline.txt 2>&1
done; done
cat $O/timeline.txt
