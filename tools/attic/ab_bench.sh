#!/bin/bash
# In-run A/B of two libraries through the whole bench line (value, tet and particle kernel):  bash tools/attic/ab_bench.sh <libA.so> <libB.so> [reps]
cd "${GRAFT_REPO_ROOT:-.}"
A=$1; B=$2; REPS=${3:-4}
for rep in $(seq $REPS); do for lib in $A $B; do
  TETSIM_HIP_LIB=$PWD/tetsim_amd/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib value %.1f ms_per_step %.4f tet %.2f us particle %.2f us' % (d['value'], d['ms_per_step'], r['kernel_us'], r['vertex_kernel_us']))"
done; done
