#!/usr/bin/env bash
# A/B of the persistent software-pipelined tet kernel (TETSIM_TET_PIPELINE = workgroups per CU; 0 = off) on a GPU box
cd $GRAFT_REPO_ROOT; OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT
for rep in 1 2; do for p in 0 4 5 6; do TETSIM_TET_PIPELINE=$p python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('pipeline=$p value %.1f ms/frame %.4f tet %.2f us vertex %.2f us frac %.3f' % (d['value'], d['ms_per_step'], r['kernel_us'], r['vertex_kernel_us'], r['frac']))"; done; done > $OUT/pipe_ab.txt 2>&1
TETSIM_TET_PIPELINE=5 timeout 600 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_polar.py -m gpu -q -k "lattice_1m or lattice_8m" 2>&1 | tail -8 > $OUT/pytest_pipe.log
cat $OUT/pipe_ab.txt; tail -4 $OUT/pytest_pipe.log
