#!/bin/bash
# Robustness batch on the GPU box: fuzz (random Delaunay meshes through every small / mid-size path against the oracle), the asynchronous
# halo choreography under stress, the persistent launches soaked, and the partition checkpoint tests repeated (a race shows up once in a while).
#   tools/robustness_batch.sh [tag] [first_seed] [seeds] [soak_seconds] [repeats]
cd "$(dirname "$0")/.."
tag=${1:-robust}; first=${2:-1000}; seeds=${3:-80}; soak=${4:-120}; reps=${5:-6}
out=gpurun_out/$tag; mkdir -p $out
timeout 900 python tools/fuzz_meshes.py $first $seeds > $out/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $out/fuzz.txt
timeout 600 python tools/stress_group.py > $out/stress_group.txt 2>&1; echo "stress rc=$?" >> $out/stress_group.txt
timeout $((soak + 240)) python tools/soak.py $soak > $out/soak.txt 2>&1; echo "soak rc=$?" >> $out/soak.txt
: > $out/repeat.txt
for i in $(seq 1 $reps); do
  timeout 600 python -m pytest tests/test_gpu_partition_state.py tests/test_gpu_frame_kernel.py -m gpu -x -q 2>&1 | tail -1 >> $out/repeat.txt
done
for f in fuzz stress_group soak; do tail -n 3 $out/$f.txt; done; cat $out/repeat.txt
