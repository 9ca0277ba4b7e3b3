cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02k; python -m tetsim_amd.build --ablation > /dev/null 2>&1
L=$PWD/tetsim_amd/libtetsim_hip_ablation.so
for rep in 1 2; do for cfg in "0 0" "8 0" "16 0" "32 0" "8 1" "16 1" "32 1" "64 1"; do set -- $cfg; TETSIM_HIP_LIB=$L TETSIM_DEBUG_STAGGER=$1 TETSIM_DEBUG_STAGGER_MAP=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stagger=$1 map=$2 value %.1f tet %.2f us frac %.3f' % (d['value'], r['kernel_us'], r['frac']))"; done; done > gpurun_out/r02k/stagger.txt 2>&1; cat gpurun_out/r02k/stagger.txt
