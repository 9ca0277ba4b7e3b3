cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02h
for f in 0 1 0 1; do TETSIM_FUSED_PARTICLE_PASS=$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('fused=$f value %.1f ms/frame %.4f kernel %s %.2f us vertex %.2f us frac %.3f' % (d['value'], d['ms_per_step'], r['kernel'], r['kernel_us'], r['vertex_kernel_us'], r['frac']))"; done > gpurun_out/r02h/fused_ab.txt 2>&1
for f in 0 1; do echo "fused=$f"; TETSIM_FUSED_PARTICLE_PASS=$f python tools/dragon_time.py 2>&1 | head -2; done >> gpurun_out/r02h/fused_ab.txt
timeout 900 python -m pytest tests/test_gpu_polar.py tests/test_gpu_polar_reference.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_gpu_random_meshes.py tests/test_gpu_skinning.py tests/test_mesh_file.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r02h/pytest.log
cat gpurun_out/r02h/fused_ab.txt; tail -12 gpurun_out/r02h/pytest.log
