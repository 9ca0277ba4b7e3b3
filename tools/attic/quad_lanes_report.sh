#!/usr/bin/env bash
# Small bodies: four lanes per tet on 64-tet tiles (pj_quad.hip) against one lane per tet on 256-tet tiles (pj_blocked.hip), on the Dragon
# (BASELINE config 2).  Run ON the GPU box:  bash tools/attic/quad_lanes_report.sh > profiles/r04_quad_lanes.txt
# 1. phase stamps of the frame kernels (ablation build, thread 0 of every tile, cycles per substep), Dragon in free fall and on the floor;
# 2. the same with the quad tiles spread over all XCDs (memory-side exchange); 3. s_sleep between a tile's store and its first look
# at the neighbours' sums (TETSIM_QUAD_POLL_DELAY, units of 64 clocks); 4. event-timed substeps of the product build, both kernels.
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TETSIM_HALO_TIMEOUT_MS=2000
python -m tetsim_amd.build --ablation > /dev/null 2>&1
for sc in "" floor; do for q in 1 0; do
  echo "== frame kernel phase stamps: TETSIM_QUAD=$q ${sc:-free fall}"
  TETSIM_QUAD=$q timeout 120 python tools/attic/frame_trace.py 20 $sc 2>&1 | grep -v "WARNING"
done; done
echo "== quad tiles over all XCDs (TETSIM_FRAME_LOCAL=0: write-through stores, loads from the memory side), floor"
TETSIM_FRAME_LOCAL=0 timeout 120 python tools/attic/frame_trace.py 20 floor 2>&1 | grep -v "WARNING\|XCC"
echo "== delay between a tile's store and its first look at the neighbours' sums (product build; floor = frame_check's Dragon 1 cm above the floor)"
for d in 0 4 8 12 16 24; do echo "delay $d x 64 clocks: floor $(TETSIM_QUAD_POLL_DELAY=$d python tools/attic/frame_check.py 2>&1 | tail -1 | sed 's/.*bit-equal/bit-equal/')  |  free fall $(TETSIM_QUAD_POLL_DELAY=$d python tools/dragon_time.py 2>&1 | sed -n 2p | cut -c66-)"; done
for d in 0 12; do echo "-- delay $d, phase stamps (floor)"; TETSIM_QUAD_POLL_DELAY=$d timeout 120 python tools/attic/frame_trace.py 20 floor 2>&1 | grep "gather\|polls"; done
echo "== per-tile timelines of two substeps on the floor (ablation build; tools/attic/frame_timeline.py): what the slowest tiles are made of"
timeout 120 python tools/attic/frame_timeline.py floor 2>&1 | grep -v "WARNING" | awk 'NR==1 || /period/ {print}' | sort -t'|' -k3 | cut -c1-8,76- | tail -64 | sort -k6 -n | awk 'NR<=8 || NR>56 {print}'
echo "== one hand-over of a tagged 8-byte value between workgroups, by how it is stored and looked at (tools/micro/handoff.hip)"
if [ -x tools/micro/bin/handoff ]; then timeout 100 tools/micro/bin/handoff; else echo "(tools/micro/bin/handoff not built)"; fi
echo "== product build, event-timed (3 repetitions each)"
for rep in 1 2 3; do for q in 1 0; do
  echo "TETSIM_QUAD=$q floor: $(TETSIM_QUAD=$q python tools/attic/frame_check.py 2>&1 | tail -1 | sed 's/.*mode/mode/')  |  free fall: $(TETSIM_QUAD=$q python tools/dragon_time.py 2>&1 | sed -n 2p | cut -c50-)"
done; done
