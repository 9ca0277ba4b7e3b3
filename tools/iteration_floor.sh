#!/usr/bin/env bash
# Iteration ablation of the PRODUCT tet kernel (run ON the GPU box; the variant libraries are built beforehand, without a GPU:
#   for n in 0 3 6; do python tools/mutant_lib.py it$n pj_lab.h '#define TETSIM_ROTATION_ITERATIONS 9' "#define TETSIM_ROTATION_ITERATIONS $n /* ABLATION OF THE PRODUCT KERNEL */"; done )
# Alternating runs of the bench line through each library; prints one line per library: the tet kernel on the floor (per-launch events,
# median of the runs), the whole substep inside the one-launch call there (its own events), and the tet kernel over the timed frames.
cd "${GRAFT_REPO_ROOT:-.}"
python - <<'PY'
import json, os, subprocess, sys, statistics
libs = [(0, "libtetsim_hip_it0.so"), (3, "libtetsim_hip_it3.so"), (6, "libtetsim_hip_it6.so"), (9, "libtetsim_hip.so")]
res = {n: [] for n, _ in libs}
for rep in range(3):
    for n, lib in libs:
        env = dict(os.environ, TETSIM_HIP_LIB=os.path.abspath(os.path.join("tetsim_amd", lib)))
        out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs", "--no-beyond-mall"],
                             env=env, capture_output=True, text=True)
        try:
            r = json.loads(out.stdout.strip().splitlines()[-1])["roofline"]
            tk = r.get("two_kernel_path", r)      # (round 6: the 1 M-tet body's dominant kernel is the whole call; the kernel pair's event figures sit below it)
            in_launch = r["on_floor"]["substep_us"] if "two_kernel_path" in r else tk["on_floor"]["in_graph"]["kernel_us_implied"]
            res[n].append((tk["on_floor"]["kernel_us"], in_launch, tk["fast_exit"]["kernel_us"], tk["vertex_kernel_us"]))
        except Exception as e:
            print("iters=%d FAILED %r %s" % (n, e, out.stderr[-300:]))
for n, _ in libs:
    if res[n]:
        med = [statistics.median(x[i] for x in res[n]) for i in range(4)]
        print("iters=%d  tet %.2f us on the floor by events (whole substep inside the one-launch call there: %.2f us; timed frames: %.2f us)  vertex %.2f us   [%d runs]" % (n, med[0], med[1], med[2], med[3], len(res[n])))
PY
