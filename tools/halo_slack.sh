#!/bin/bash
# How much wire latency the halo choreography hides: one interior rank in loopback (tools/loopback_rank.py), RCCL transfer vs
# peer-to-peer stores, with 0 / 10 / 20 us of injected delay (TETSIM_DEBUG_LOOPBACK_DELAY_US).  Writes $1 (default gpurun_out/halo_slack.txt).
# DELAYS="0 20" BND="1 0" select the sweep (BND: TETSIM_HALO_BND_IN_TILES, boundary particles finished by the halo-side tiles or by a kernel of their own).
out=${1:-gpurun_out/halo_slack.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
export TETSIM_HALO_TIMEOUT_MS=5000 LOOPBACK_CALLS=40 LOOPBACK_REPS=3
first=1
for bnd in ${BND:-1}; do
 for p2p in "" 1; do
  for d in ${DELAYS:-0 10 20}; do
    echo "== transport: $([ -n "$p2p" ] && echo peer-to-peer || echo rccl)   boundary particles in the halo-side tiles: $bnd   injected delay: $d us" >> "$out"
    TETSIM_HALO_BND_IN_TILES=$bnd LOOPBACK_P2P=$p2p TETSIM_DEBUG_LOOPBACK_DELAY_US=$d LOOPBACK_SKIP_MONO=$([ $first = 1 ] || echo 1) timeout 300 python tools/loopback_rank.py 2>&1 | grep "wall" >> "$out"
    first=0
  done
 done
done
cat "$out"
