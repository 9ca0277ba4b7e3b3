#!/usr/bin/env bash
set -u
TAG=${1:-r02e}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"
python -m tetsim_amd.build > "$OUT/build.log" 2>&1
echo "== polar: previous build vs load-order fix" > "$OUT/ab.txt"
timeout 600 python tools/ab_lib.py tetsim_amd/libtetsim_hip_prev.so tetsim_amd/libtetsim_hip.so >> "$OUT/ab.txt" 2>&1
for lib in tetsim_amd/libtetsim_hip_prev.so tetsim_amd/libtetsim_hip.so; do
  echo "== neo-hookean, $lib" >> "$OUT/nh.txt"
  TETSIM_HIP_LIB=$ROOT/$lib timeout 300 python tools/nh_time.py >> "$OUT/nh.txt" 2>&1
  TETSIM_HIP_LIB=$ROOT/$lib timeout 300 python tools/dragon_time.py >> "$OUT/nh.txt" 2>&1
done
timeout 900 python -m pytest tests/test_gpu_neohookean.py tests/test_gpu_polar.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -15 > "$OUT/pytest.log"
cat "$OUT/ab.txt" "$OUT/nh.txt"; tail -5 "$OUT/pytest.log"
