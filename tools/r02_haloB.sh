#!/bin/bash
# r02n: the halo choreography with the boundary particles on the halo queue -- liveness/determinism, partition parity, and how much
# wire latency the loopback rank hides (TETSIM_DEBUG_LOOPBACK_DELAY_US adds an idle kernel behind every transfer)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r02n; O=gpurun_out/r02n
[ -n "${SKIP_TESTS:-}" ] || timeout 1500 python -m pytest tests -x -q -m gpu -k "partition or rccl or group or halo or loopback or multi" > $O/halo_tests.log 2>&1; echo "halo tests rc=$?" >> $O/halo_tests.log
tail -5 $O/halo_tests.log
# A = the previous choreography (boundary particles on the main queue, transfer behind `wait V`), built as libtetsim_hip_haloA.so
for lib in libtetsim_hip.so libtetsim_hip_haloA.so; do for g in 1 0; do for d in 0 10 20 30 40; do
  [ -f tetsim_amd/$lib ] || continue
  echo "== $lib TETSIM_HALO_GRAPH=$g delay=$d us" >> $O/loopback.log
  TETSIM_HIP_LIB=$PWD/tetsim_amd/$lib TETSIM_HALO_GRAPH=$g TETSIM_DEBUG_LOOPBACK_DELAY_US=$d timeout 300 python tools/loopback_rank.py 2>&1 | grep 'wall' | tail -4 >> $O/loopback.log
done; done; done
cat $O/loopback.log
