# first-round stagger ablation of the tet kernel (ablation library; TETSIM_DEBUG_STAGGER / _MAP, pj_blocked.hip)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${STAGGER_OUT:-r02k}; mkdir -p $O; python -m tetsim_amd.build --ablation > /dev/null 2>&1
L=$PWD/tetsim_amd/libtetsim_hip_ablation.so
for rep in 1 2; do for cfg in ${STAGGER_CFGS:-0:0 8:0 16:0 32:0 8:1 16:1 32:1 64:1}; do st=${cfg%%:*}; mp=${cfg##*:}
  TETSIM_HIP_LIB=$L TETSIM_DEBUG_STAGGER=$st TETSIM_DEBUG_STAGGER_MAP=$mp python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stagger=$st map=$mp value %.1f tet %.2f us frac %.3f' % (d['value'], r['kernel_us'], r['frac']))"; done; done > $O/stagger.txt 2>&1; cat $O/stagger.txt
