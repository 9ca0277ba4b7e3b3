#!/bin/bash
# In-run A/B of environment knobs through the whole bench line: tools/ab_env.sh OUT "NAME=VALUE ..." "NAME=VALUE ..." ...  ("-" = no knob).
# Three alternating rounds of bench.py --steps 40 (headline workload, no CPU baseline / other configs).  TETSIM_HIP_LIB may select a library per variant.
out=$1; shift
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
    line=$(env $envs python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-beyond-mall 2>/dev/null | tail -1)
    python - "$v" "$line" >> "$out" <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line); r = d["roofline"]
    print("%-44s value %8.1f  ms/frame %.4f  tet %.2f us  particle %.2f us  frac %.4f  substep frac %.4f" % (v, d["value"], d["ms_per_step"], r["kernel_us"], r["vertex_kernel_us"], r["frac"], r["substep_frac"]))
except Exception as e:
    print("%-44s FAILED %s" % (v, str(e)[:80]))
PY
  done
done
cat "$out"
