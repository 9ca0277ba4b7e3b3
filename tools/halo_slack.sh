#!/bin/bash
# How much wire latency the halo choreography hides: one interior rank in loopback (tools/loopback_rank.py), RCCL transfer vs
# peer-to-peer stores, with 0 / 10 / 20 us of injected delay (TETSIM_DEBUG_LOOPBACK_DELAY_US).  Writes $1 (default gpurun_out/halo_slack.txt).
# DELAYS="0 20" MODES="rccl p2p deep" select the sweep (deep = peer-to-peer stores over a two-layer ghost region, ghosts every other substep).
out=${1:-gpurun_out/halo_slack.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
export TETSIM_HALO_TIMEOUT_MS=5000 LOOPBACK_CALLS=40 LOOPBACK_REPS=3
first=1
for mode in ${MODES:-rccl p2p deep}; do
  for d in ${DELAYS:-0 10 20}; do
    echo "== transport: $mode   injected delay: $d us" >> "$out"
    LOOPBACK_P2P=$([ $mode = p2p ] && echo 1) LOOPBACK_DEEP=$([ $mode = deep ] && echo 1) TETSIM_DEBUG_LOOPBACK_DELAY_US=$d LOOPBACK_SKIP_MONO=$([ $first = 1 ] || echo 1) timeout 300 python tools/loopback_rank.py 2>&1 | grep "wall\|tet kernel" >> "$out"
    first=0
  done
done
cat "$out"
