"""The CPU leg of the bench line (`cpu_baseline`): the oracle's restatement of the SAME algorithm timed on this host's cores, after the
timed region.  Only this file (and other_configs' node leg) touches oracle/."""
import json
import os
import time

import numpy as np

from .common import DT, PP, ROOT, SUBSTEPS

def cpu_budget():
    """What this process may actually use of the host: the CPUs it may be scheduled on (sched_getaffinity) and the cgroup's CPU-time quota
    (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us) in units of CPUs -- a container that SEES 256 hardware threads but is
    throttled to 16 CPUs' worth of time gets slower, not faster, beyond 16 threads."""
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    quota, src = None, None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            quota, src = float(q) / float(period), "cgroup v2 cpu.max"
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0:
                quota, src = q / period, "cgroup v1 cpu.cfs_quota_us"
        except Exception:
            pass
    return avail, quota, src

def cpu_baseline(verts, tets):
    """Time the CPU restatement of the SAME algorithm (oracle section G) on this host by BASELINE.md 4.2's protocol: same
    lattice, parameters and dt as the GPU run; 3 repetitions and their MEDIAN at 1 thread (like-for-like with the reference's
    single JS thread), at the host's CPU BUDGET (the cgroup quota if there is one, else the CPUs the process may run on, capped at 64:
    beyond one socket's worth of cores the port stops scaling), at half and at twice that (OpenMP over tets / particles).  Every
    thread count is reported (`by_threads`), `value` / `cores` name the fastest median.  Bounded to ~25 s of CPU work."""
    from oracle import OraclePJ, set_threads
    avail, quota, quota_src = cpu_budget()
    budget = max(1, min(avail, int(round(quota)) if quota else min(avail, 64)))
    o = OraclePJ(verts, tets, PP, slot_quirk=True)
    REPS = 3

    def rate(threads, budget_s):
        """REPS repetitions of n substeps each (n from a one-substep probe so that a repetition lasts ~budget_s / REPS)."""
        set_threads(threads)
        o.simulate(DT, PP)  # warm (page faults, thread pool)
        t0 = time.perf_counter()
        o.simulate(DT, PP)
        t1 = time.perf_counter() - t0
        n = int(max(1, min(SUBSTEPS, budget_s / REPS / max(t1, 1e-3))))
        rates = []
        for _ in range(REPS):
            t0 = time.perf_counter()
            for _ in range(n):
                o.simulate(DT, PP)
            rates.append(len(tets) * n / (time.perf_counter() - t0) / 1e6)
        rates.sort()
        return {"median": round(rates[REPS // 2], 3), "min": round(rates[0], 3), "max": round(rates[-1], 3), "reps": REPS, "substeps_per_rep": n}

    by_threads = {}
    for th, seconds in ((1, 4.0), (max(1, budget // 2), 2.5), (budget, 3.0), (min(avail, 2 * budget), 2.5)):
        th = min(th, avail)
        if str(th) not in by_threads:
            by_threads[str(th)] = rate(th, seconds)
    set_threads(1)
    cores = max(by_threads, key=lambda k: by_threads[k]["median"])
    # the reference's own CPU solver is the sequential Neo-Hookean Gauss-Seidel of Softbody.js (BASELINE config 1); its
    # restatement (oracle section A, bit-exact with Softbody.js) on ONE core of this host, same lattice, for orientation
    from oracle import OracleNH
    nh = OracleNH(verts, tets, PP)
    nh.simulate(DT, PP)
    nh_rates = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        nh.simulate(DT, PP)
        nh.simulate(DT, PP)
        nh_rates.append(2 * len(tets) / (time.perf_counter() - t0) / 1e6)
    nh_rates.sort()
    # ... and the same algorithm in JavaScript under node (oracle/nh_port.js, bit-exact with Softbody.js on the golden
    # vectors): the reference's design point -- one JS thread -- on this host
    js = None
    import shutil
    import subprocess
    import tempfile
    node = shutil.which("node")
    if node:
        try:
            with tempfile.TemporaryDirectory() as tmp:
                np.ascontiguousarray(verts, dtype="<f4").tofile(os.path.join(tmp, "v.f32"))
                np.ascontiguousarray(tets, dtype="<i4").tofile(os.path.join(tmp, "t.i32"))
                r = subprocess.run([node, os.path.join(ROOT, "oracle", "nh_port.js"), "--verts", os.path.join(tmp, "v.f32"), "--tets",
                                    os.path.join(tmp, "t.i32"), "--substeps", "3", "--reps", str(REPS), "--warmup", "1", "--per-frame", str(SUBSTEPS)],
                                   capture_output=True, text=True, timeout=300)
            jr = json.loads(r.stdout)
            js = {"value": round(jr["m_tet_solves_per_s"], 3), "unit": "M tet-solves/s", "cores": 1, "kind": "port",
                  "min": round(min(jr["rates"]), 3), "max": round(max(jr["rates"]), 3), "reps": jr["reps"],
                  "sample": "median of %d repetitions of 3 substeps of the same lattice after 1 warm-up, oracle/nh_port.js under node %s" % (jr["reps"], jr["node"])}
        except Exception as e:  # the JS leg is optional: node may be absent or too old
            js = {"error": str(e)[:200]}
    best = by_threads[cores]
    res = {"value": best["median"], "unit": "M tet-solves/s", "cores": int(cores), "kind": "port",
           "by_threads": by_threads,
           "softbody_js_algorithm_node_1thread": js,
           "softbody_js_algorithm_1core": {"value": round(nh_rates[REPS // 2], 3), "min": round(nh_rates[0], 3), "max": round(nh_rates[-1], 3), "reps": REPS,
                                           "unit": "M tet-solves/s", "cores": 1, "kind": "port",
                                           "sample": "median of %d repetitions of 2 substeps of the same lattice, sequential Neo-Hookean Gauss-Seidel (oracle section A)" % REPS},
           "sample": "median of %d repetitions of %d substeps of the same %d-tet lattice, same parameters and dt as the GPU run (oracle/tetsim_oracle.c "
                     "section G, gcc -O2 + OpenMP over tets/particles); thread counts 1, half the CPU budget, the budget and twice the budget are in "
                     "by_threads, value = the fastest median" % (REPS, best["substeps_per_rep"], len(tets)),
           "value_1core": by_threads["1"]["median"], "host_cpus_available": avail,
           "cpu_budget": {"cpus": budget, "cgroup_quota_cpus": round(quota, 2) if quota else None,
                          "source": quota_src or ("no cgroup CPU quota: the CPUs this process may run on (sched_getaffinity)" + (", capped at 64" if avail > 64 else ""))}}
    try:
        with open("/proc/cpuinfo") as f:
            res["cpu"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return res
