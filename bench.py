#!/usr/bin/env python3
"""bench.py -- headline benchmark: M tet-solves/s on the 1 M-tet Kuhn lattice, polar-decomposition Jacobi,
20 substeps per frame, on N MI355X (BASELINE.json metric; SURVEY.md §8(d) config 3, config 5 shape for N>1).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: spawns its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one animation frame of the reference's driver loop (main.js:79-84): 20 substeps of the hot path
over the whole body, issued as ONE tetsim_step_n call (one HIP-graph launch).  All state is resident in HBM
before the timed region; nothing is read back inside it.  N > 1: weak scaling -- the lattice is stacked to
55 x 55 x 55N cells, slab-partitioned along z (one slab, ~1 M tets, per GPU), ghost-vertex positions cross
xGMI once per substep through RCCL send/recv issued by libtetsim_hip itself.

The JSON line carries two extra objects:
  roofline      the dominant kernel (pj_tet_kernel, P3+P4 of the reference) against HBM peak.  `achieved` =
                algorithmic bytes per launch / mean launch duration measured here with HIP events on the
                handle's own stream.  Bytes per tet are SURVEY.md §8(d)'s tet row (148 B) -- see DESIGN.md.
  cpu_baseline  the CPU restatement (oracle/, "port") of the same algorithm on this host's cores, bounded sample.
and, outside the timed region:
  library        what was loaded: ABI, source / kernel hashes, "ablation": false for the product build, debug env knobs set
  other_configs  (N = 1) BASELINE configs 1, 2 and 4 on the same box, bounded to a few seconds
  multi_gpu      (N > 1) ranks RCCL itself reports, per-rank step time min/max, halo bytes, host enqueue time per substep;
                 the run FAILS if RCCL's rank count differs from --gpus
  config5_strong (N = 8, or --config5 on) BASELINE config 5 literally: the 110^3-cell lattice cut into N slabs
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SUBSTEPS = 20
CELLS = 55
PP = dict(gravity=-9.81, timeScale=1.0, timeStep=1.0 / 60.0, numSubsteps=SUBSTEPS, friction=1000.0,
          density=1000.0, devCompliance=1.0 / 100000.0, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])
DT = (PP["timeScale"] * PP["timeStep"]) / PP["numSubsteps"]  # main.js:79

# SURVEY.md §8(d), reference formulation (world-space lastRest carried forward), per tet per substep:
# idx 16 R + lastRest 48 R + 48 W + quat 16 R + 16 W + restVol 4 R.
TET_KERNEL_BYTES = 148.0
VERTEX_BYTES = 144.0           # per particle per substep (integrate/accumulate/finalize rows)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


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


GOLD = os.path.join(ROOT, "tests", "golden")


def other_configs(steps=20, warmup=5):
    """BASELINE configs 1, 2 and 4 on this box, outside the timed region, a few seconds in total.  (Config 3 is the line itself,
    config 5 needs N > 1.)  Same metric everywhere: M tet-solves/s = tets x substeps / wall."""
    import shutil
    import subprocess
    from tetsim_amd import SoftBodyHIP, make_lattice
    out = {}
    dv = np.fromfile(os.path.join(GOLD, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3)
    dtets = np.fromfile(os.path.join(GOLD, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)

    def hip_rate(v, t, n_sub, frames, **kw):
        body = SoftBodyHIP(v, t, None, dict(PP), **kw)
        dt = (PP["timeScale"] * PP["timeStep"]) / n_sub
        body.simulateSubsteps(n_sub, dt, PP)
        body.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            body.simulateSubsteps(n_sub, dt, PP)
        body.sync()
        el = time.perf_counter() - t0
        levels = body.info.num_levels
        mode = int(body.info.fused_particle_pass)   # 0: tet + particle kernel per substep; 1: one fused kernel per substep; 2: one persistent kernel per frame
        body.close()
        return {"value": round(len(t) * n_sub * frames / el / 1e6, 2), "unit": "M tet-solves/s", "ms_per_frame": round(el / frames * 1e3, 4),
                "us_per_substep": round(el / frames / n_sub * 1e6, 2), "frames": frames,
                "launches_per_substep": (levels + 1) if levels else {0: 2, 1: 1, 2: round(1.0 / n_sub, 3)}[mode]}

    # config 1: Dragon, the reference's CPU solver (Neo-Hookean Gauss-Seidel), 10 substeps per frame
    c1 = {"workload": "Dragon (%d tets, %d particles), Neo-Hookean XPBD Gauss-Seidel, 10 substeps/frame" % (len(dtets), len(dv))}
    node = shutil.which("node")
    if node:
        try:
            r = subprocess.run([node, os.path.join(ROOT, "oracle", "nh_port.js"), "--verts", os.path.join(GOLD, "dragon_verts.f32"), "--tets",
                                os.path.join(GOLD, "dragon_tets.i32"), "--substeps", "400", "--warmup", "100", "--per-frame", "10"],
                               capture_output=True, text=True, timeout=120)
            jr = json.loads(r.stdout)
            c1["softbody_js_algorithm_node_1thread"] = {"value": round(jr["m_tet_solves_per_s"], 3), "unit": "M tet-solves/s", "cores": 1, "kind": "port",
                                                        "sample": "400 substeps after 100 warm-up, oracle/nh_port.js under node " + jr["node"]}
        except Exception as e:  # node is optional
            c1["softbody_js_algorithm_node_1thread"] = {"error": str(e)[:200]}
    c1["hip_original_order_precise"] = dict(hip_rate(dv, dtets, 10, 10, solver="neohookean", precision="precise", order="original"),
                                            note="bit-exact with Softbody.js in the caller's tet order (703 dependency levels)")
    c1["hip_coloured_precise"] = dict(hip_rate(dv, dtets, 10, 100, solver="neohookean", precision="precise", order="coloured"),
                                      note="bit-exact with Softbody.js fed tetIds[tetsim_get_tet_order()]")
    out["config1_dragon_neohookean_cpu_path"] = c1
    # config 2: Dragon, polar-decomposition Jacobi, f32, 20 substeps per frame
    out["config2_dragon_polar_jacobi"] = {
        "workload": "Dragon, polar-decomposition Jacobi, 20 substeps/frame, one graph launch per frame (FAST: ONE persistent kernel per frame, "
                    "every tile's workgroup resident for the 20 substeps; PRECISE: a tet and a particle kernel per substep)",
        "fast": hip_rate(dv, dtets, 20, 400, solver="polar", precision="fast"),
        "precise": hip_rate(dv, dtets, 20, 200, solver="polar", precision="precise")}
    # config 3 once more with the REFERENCE's rotation-exit threshold (TETSIM_FLAG_REF_ROTATION_EXIT: |omega| < 1e-9, i.e. all nine iterations
    # in f32, SoftbodyGPU.js:131) -- the headline's FAST default ends a tet's correction iterations below 1e-6 rad.  Same lattice, same
    # protocol as the headline (warm-up + timed frames from rest); what the threshold is worth depends on the phase of the fall.
    lv, lt = make_lattice(CELLS)
    body = SoftBodyHIP(lv, lt, None, dict(PP), solver="polar", precision="fast", ref_rotation_exit=True)
    for _ in range(warmup):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    el = time.perf_counter() - t0
    body.close()
    out["config3_reference_threshold"] = {
        "workload": "the headline's lattice and frames (%d warm-up + %d timed, from rest) with rotation_exit = the reference's |omega| < 1e-9" % (warmup, steps),
        "value": round(len(lt) * SUBSTEPS * steps / el / 1e6, 1), "unit": "M tet-solves/s", "ms_per_step": round(el / steps * 1e3, 4)}
    # config 4: Neo-Hookean Gauss-Seidel on the 1 M-tet lattice + convergence against Jacobi (dropped 2 cm onto the floor)
    v, t = make_lattice(CELLS, y0=0.02)
    Dm_inv = np.linalg.inv((v[t[:, 1:]] - v[t[:, :1]]).astype(np.float64).transpose(0, 2, 1))

    def vol_residual(pos):   # mean |det F - 1|: the reference's volError analogue (Softbody.js:163)
        F = (pos[t[:, 1:]] - pos[t[:, :1]]).astype(np.float64).transpose(0, 2, 1) @ Dm_inv
        return float(np.abs(np.linalg.det(F) - 1.0).mean())

    c4 = {"workload": "Kuhn-6 lattice %d^3 cells (%d tets) dropped 2 cm onto the floor, %d substeps/frame" % (CELLS, len(t), SUBSTEPS),
          "residual": "mean |det F - 1| after 1 / 5 / 30 frames, evaluated on the host in f64 from the returned positions"}
    for key, kw in (("neohookean_clustered_gs_fast", dict(solver="neohookean", precision="fast", order="clustered")),
                    ("neohookean_clustered_gs_precise", dict(solver="neohookean", precision="precise", order="clustered")),
                    ("polar_jacobi_fast", dict(solver="polar", precision="fast"))):
        body = SoftBodyHIP(v, t, None, dict(PP), **kw)
        snaps, done = [], 0
        for frames in (1, 5, 30):
            for _ in range(frames - done):
                body.simulateSubsteps(SUBSTEPS, DT, PP)
            done = frames
            snaps.append(body.pos.copy())   # (the f64 residual of 1 M tets takes the host ~0.3 s: evaluated AFTER the timed frames, so that they do not start from an idle device)
        body.sync()
        t0 = time.perf_counter()   # the rate: 20 more frames of the same body (resting on the floor by now), graph already built
        for _ in range(20):
            body.simulateSubsteps(SUBSTEPS, DT, PP)
        body.sync()
        el = time.perf_counter() - t0
        res = [float("%.3e" % vol_residual(p)) for p in snaps]
        c4[key] = {"value": round(len(t) * SUBSTEPS * 20 / el / 1e6, 1), "unit": "M tet-solves/s", "ms_per_frame": round(el / 20 * 1e3, 4),
                   "mean_abs_detF_minus_1_after_1_5_30_frames": res,
                   "launches_per_substep": (body.info.num_levels + (0 if body.info.fused_particle_pass else 1)) if body.info.num_levels else 2}
        body.close()
    out["config4_lattice_1m_neohookean_gs_vs_jacobi"] = c4
    return out


REAL_STDOUT_FD = None   # the process's real stdout while fd 1 is routed to stderr (HeadlineGuard prints there)


class stdout_to_stderr:
    """Route fd 1 to fd 2 while native libraries initialise (RCCL prints a version banner on stdout): rank 0's stdout
    must carry exactly one JSON line."""

    def __enter__(self):
        global REAL_STDOUT_FD
        sys.stdout.flush()
        self._saved = os.dup(1)
        REAL_STDOUT_FD = self._saved
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:    # RCCL's banner is printf'ed: with stdout a pipe it sits in C's buffer and would come out at exit, behind the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        global REAL_STDOUT_FD
        os.dup2(self._saved, 1)
        os.close(self._saved)
        REAL_STDOUT_FD = None


class HeadlineGuard:
    """The optional legs that follow the headline at N > 1 (the peer-to-peer halo check, BASELINE config 5) are collective: a rank
    that fails alone leaves the others inside a barrier.  They must never cost the headline.  Once the headline line is complete it
    is armed with a budget; if the legs are not done by then, rank 0 prints the headline as it stands (plus a note saying what was
    cut short) on the real stdout and every rank leaves at once -- exit code 0, one JSON line, as the contract wants."""

    def __init__(self):
        self._timer = None

    def arm(self, line, seconds, what):
        import threading
        self.disarm()

        def fire():
            try:
                if line is not None:
                    line.setdefault("notes", []).append("%s did not finish within %d s and was cut short; the headline above is complete" % (what, seconds))
                    data = (json.dumps(line) + "\n").encode()
                    fd = REAL_STDOUT_FD if REAL_STDOUT_FD is not None else 1
                    while data:
                        data = data[os.write(fd, data):]
            finally:
                os._exit(0)

        self._timer = threading.Timer(seconds, fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None


GUARD = HeadlineGuard()


class TorchRanks:
    """One process per GPU (the driver's launch): torch.distributed over RCCL for rendezvous, barrier and the max over ranks.
    The halo traffic itself does not go through torch: libtetsim_hip owns its RCCL communicator."""

    def __init__(self, local_rank, rank=0, world=1):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")     # --force-dist without a launcher: a one-rank rendezvous with itself
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port())
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def broadcast_bytes(self, data, n):           # rank 0's `data` (n bytes) to everyone
        t = self.torch.zeros(n, dtype=self.torch.uint8, device="cuda")
        if data is not None:
            t.copy_(self.torch.tensor(list(data), dtype=self.torch.uint8))
        self.dist.broadcast(t, src=0)
        return bytes(t.cpu().tolist())

    def all_gather_bytes(self, data, n):          # every rank's `data` (n bytes), in rank order
        t = self.torch.tensor(list(data), dtype=self.torch.uint8, device="cuda")
        out = self.torch.empty(n * self.dist.get_world_size(), dtype=self.torch.uint8, device="cuda")
        self.dist.all_gather_into_tensor(out, t)
        flat = bytes(out.cpu().tolist())
        return [flat[i * n:(i + 1) * n] for i in range(self.dist.get_world_size())]

    def barrier(self):
        self.torch.cuda.synchronize()
        self.dist.barrier()

    def max_float(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_float(self, x):
        return -self.max_float(-x)

    def close(self):
        self.dist.destroy_process_group()


class ThreadRanks:
    """--fake-ranks N (development / tests on a ONE-GPU box): the N ranks are host threads of this process, all on device 0,
    and librccl is the strict test double of tests/mock_rccl (TETSIM_RCCL_LIB).  Same code path as the real launch from
    `run()` down; only this adapter differs."""

    def __init__(self, shared, rank):
        self.s, self.rank = shared, rank

    def broadcast_bytes(self, data, n):
        if data is not None:
            self.s["bytes"] = bytes(data)
        self.s["barrier"].wait()
        out = self.s["bytes"]
        self.s["barrier"].wait()
        return out

    def all_gather_bytes(self, data, n):
        self.s.setdefault("gather", [None] * len(self.s["vals"]))[self.rank] = bytes(data)
        self.s["barrier"].wait()
        out = list(self.s["gather"])
        self.s["barrier"].wait()
        return out

    def barrier(self):
        self.s["barrier"].wait()

    def max_float(self, x):
        self.s["vals"][self.rank] = x
        self.s["barrier"].wait()
        out = max(self.s["vals"])
        self.s["barrier"].wait()
        return out

    def min_float(self, x):
        return -self.max_float(-x)

    def close(self):
        pass


def visible_devices():
    """HIP devices this process can see (torch is the plumbing the ranks use anyway; no context is created by the count)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:  # noqa: BLE001
        return 0


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv, worker=None, devices=None, limit_s=None, out=None, err=None):
    """`python bench.py --gpus N` without a launcher: be the launcher.  Starts N rank processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, one device each through LOCAL_RANK, a free rendezvous port on 127.0.0.1), forwards rank 0's
    stdout -- the ONE JSON line -- to this process's stdout and every other rank's stdout to stderr, and returns the exit code:
    0 if every rank exited 0, otherwise the first non-zero code seen (the surviving ranks are terminated by PID, never by pattern).
    A rank that dies takes the launch down at once instead of leaving its peers in a collective until the watchdog fires.

    worker / devices / limit_s / out / err are for the CPU test of this logic (a stub worker, a pretended device count)."""
    import subprocess
    import threading
    out = out or sys.stdout
    err = err or sys.stderr
    have = visible_devices() if devices is None else devices
    if have < n:
        err.write("bench.py: %d devices requested, %d visible\n" % (n, have))
        return 2
    worker = worker or [sys.executable, os.path.abspath(__file__)]
    limit_s = limit_s if limit_s is not None else float(os.environ.get("TETSIM_BENCH_WATCHDOG_S", "600")) + 30.0
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs, pumps = [], []

    def pump(src, dst, lock=threading.Lock()):
        for line in iter(src.readline, ""):
            with lock:
                dst.write(line)
                dst.flush()

    for r in range(n):
        p = subprocess.Popen(worker + list(argv), env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=None,
                             text=True, bufsize=1)
        procs.append(p)
        th = threading.Thread(target=pump, args=(p.stdout, out if r == 0 else err), daemon=True)
        th.start()
        pumps.append(th)
    deadline = time.monotonic() + limit_s
    code, live = 0, set(range(n))
    while live and code == 0:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    err.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, rc))
        if time.monotonic() > deadline and live:
            err.write("bench.py: ranks %s still running after %.0f s; stopping them\n" % (sorted(live), limit_s))
            code = 124
        if live and code == 0:
            time.sleep(0.05)
    for r in live:                      # only on failure: the ranks that are still up
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=10)
        except subprocess.TimeoutExpired:
            procs[r].kill()
            procs[r].wait()
    for th in pumps:
        th.join(timeout=5)
    return code


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 frames = 4000 substeps = 0.17 s timed: past the clock ramp of a short run
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fast", choices=["fast", "precise"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solver", default="polar", choices=["polar", "neohookean"],
                    help="polar (default: the headline, BASELINE configs 3/5) or neohookean (config 4: coloured Gauss-Seidel on the same "
                         "lattice; single GPU only -- it does not partition)")
    ap.add_argument("--order", default="clustered", choices=["coloured", "clustered"], help="--solver neohookean: Gauss-Seidel schedule")
    ap.add_argument("--constant-rest-shape", action="store_true",
                    help="opt-in TETSIM_FLAG_CONSTANT_REST_SHAPE formulation (NOT the headline: 100 instead of 148 algorithmic B/tet)")
    ap.add_argument("--cells", type=int, default=CELLS, help="lattice cells per side (default 55 = the 1 M-tet headline; 110 = 8 M tets)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver measures): cells^2 x (cells*N) lattice, one slab per GPU; strong: the cells^3 "
                         "lattice split into N slabs (BASELINE config 5 is --scaling strong --cells 110)")
    ap.add_argument("--profile-ranks", action="store_true",
                    help="N > 1: after the timed region every rank runs 60 more substeps with per-kernel events and rank 0 reports "
                         "its interior tet kernel in `roofline` (default at N > 1: whole-substep figures only)")
    ap.add_argument("--config5", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: append BASELINE config 5 literally (the --config5-cells^3 lattice cut into N slabs, strong scaling) as "
                         "`config5_strong` of the same JSON line; auto = when N == 8")
    ap.add_argument("--config5-cells", type=int, default=110, help="cells per side of the config-5 body (110 = 7,986,000 tets)")
    ap.add_argument("--no-beyond-mall", action="store_true", help="N = 1: skip `roofline.beyond_mall` (the 8 M-tet body on this GPU)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the `other_configs` object (BASELINE configs 1, 2, 4)")
    ap.add_argument("--halo", default="rccl", choices=["rccl", "p2p", "deep"],
                    help="N > 1: transport of the per-substep ghost exchange of the HEADLINE run -- rccl (default: grouped ncclSend/ncclRecv), p2p "
                         "(the boundary-particle kernel stores straight into the neighbours' ghost ranges, mapped through HIP IPC) or deep (p2p over a "
                         "two-layer ghost region: ghosts cross every other substep, TETSIM_FLAG_DEEP_GHOSTS)")
    ap.add_argument("--p2p-check", default="auto", choices=["auto", "on", "off"],
                    help="N > 1 with --halo rccl: afterwards repeat the run on a fresh body with the peer-to-peer halo and report its rate and whether its "
                         "positions equal the RCCL run's bit for bit (`multi_gpu.p2p_halo`); auto = on, except with --fake-ranks")
    ap.add_argument("--headline-halo", default="best", choices=["best", "rccl"],
                    help="N > 1 with --halo rccl and the peer-to-peer check: best (default) = report the faster transport as the headline IF the "
                         "peer-to-peer run was validated in this run (bit-equal positions, same frames, same protocol), with the RCCL figures beside it "
                         "in multi_gpu.rccl_halo; rccl = the RCCL run is the headline whatever the check says")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (smoke test)")
    ap.add_argument("--self-spawn", action="store_true",
                    help="launch the rank processes from this process even when --gpus is 1 (what a plain `python bench.py --gpus N`, "
                         "N > 1, does by itself when no launcher set WORLD_SIZE)")
    ap.add_argument("--fake-ranks", type=int, default=0,
                    help="development: run N ranks as threads on ONE GPU against the RCCL test double (needs TETSIM_RCCL_LIB=tests/mock_rccl/...)")
    return ap.parse_args()


def main():
    # a rank that is stuck in a collective would otherwise hang the whole launch until someone's outer limit fires
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("TETSIM_BENCH_WATCHDOG_S", "600")), exit=True)
    args = parse_args()
    if args.fake_ranks > 1:
        import threading
        if "mock_rccl" not in os.environ.get("TETSIM_RCCL_LIB", ""):
            raise SystemExit("--fake-ranks needs TETSIM_RCCL_LIB to point at the test double (tests/mock_rccl/libmock_rccl.so)")
        os.environ["TETSIM_HALO_GRAPH"] = "0"   # the test double rendezvouses on the host: not capturable
        n = args.fake_ranks
        shared = {"barrier": threading.Barrier(n), "vals": [0.0] * n, "bytes": None}
        results, errors = [None] * n, []

        def rank_main(r):
            try:
                out, body = run(args, r, n, 0, ThreadRanks(shared, r))
                shared["barrier"].wait()
                body.close()
                results[r] = out
            except BaseException as e:  # noqa: BLE001
                errors.append("rank %d: %r" % (r, e))
                shared["barrier"].abort()

        with stdout_to_stderr():
            threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
        if errors:
            raise SystemExit("; ".join(errors))
        print(json.dumps(results[0]), flush=True)
        faulthandler.cancel_dump_traceback_later()
        return

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_spawn):
        # a plain `python bench.py --gpus N` (the driver's N = 1 command with N changed): no launcher gave this process a rank, so
        # it becomes the launcher -- one rank process per GPU, rank 0's JSON line forwarded (the torchrun launch keeps working:
        # there WORLD_SIZE is set and this branch is not taken)
        faulthandler.cancel_dump_traceback_later()
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.self_spawn:
        args.force_dist = True      # a spawned rank always takes the torch.distributed path, also when it is the only one
    with stdout_to_stderr():
        ranks = TorchRanks(local_rank, rank, world) if (world > 1 or args.force_dist) else None
        out, body = run(args, rank, world, local_rank, ranks)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if ranks is not None:
        with stdout_to_stderr():
            ranks.barrier()
            body.close()
            ranks.close()
    faulthandler.cancel_dump_traceback_later()


def run_neohookean(args, verts, tets, device):
    """BASELINE config 4 on request (`--solver neohookean`): Neo-Hookean XPBD Gauss-Seidel (Softbody.js's algorithm, coloured or
    clustered schedule) on the same lattice and metric.  PRECISE reproduces Softbody.js bit for bit on the permuted tet order."""
    from tetsim_amd import SoftBodyHIP
    body = SoftBodyHIP(verts, tets, None, dict(PP), solver="neohookean", precision=args.precision, order=args.order, device=device)
    for _ in range(args.warmup):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    elapsed = time.perf_counter() - t0
    if not np.isfinite(body.pos).all():
        raise SystemExit("non-finite positions after the timed region")
    value = len(tets) * SUBSTEPS * args.steps / elapsed / 1e6
    b_alg = 56.0 + 124.0 * len(verts) / len(tets)   # idx 16 + invRestPose 36 + invRestVolume 4; 124 B per particle
    agg = b_alg * value * 1e6 / 1e9
    pr = body.profile(SUBSTEPS * 3, DT, PP)
    out = {
        "metric": "tet_solves_per_sec", "value": round(value, 1), "unit": "M tet-solves/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if args.precision == "precise" else "f32", "data": "synthetic",
        "config": {"workload": "Kuhn-6 cube lattice %dx%dx%d cells (%d tets, %d particles), Neo-Hookean XPBD Gauss-Seidel (%s schedule, "
                               "%d launches per substep), %d substeps/frame, dt=1/1200 s" % (args.cells, args.cells, args.cells, len(tets), len(verts), args.order,
                                                                                           body.info.num_levels, SUBSTEPS),
                   "solver": "neohookean_gs", "arithmetic": args.precision, "order": args.order, "substeps_per_step": SUBSTEPS,
                   "tets": len(tets), "particles": len(verts), "parallelism": "single GPU"},
        # the bound of this solver is its dependency chain (launches x (launch + round trips) + sequential tet solves);
        # the HBM figure is reported because the contract asks for one
        "roofline": {"bound": "hbm", "kernel": "whole substep (Gauss-Seidel sweep + particle pass)", "achieved": round(agg, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(agg / HBM_PEAK_GBS, 4), "traffic": None,
                     "sweep_us_per_substep": round(pr["tet_ms"] / pr["substeps"] * 1e3, 2),
                     "particle_us_per_substep": round(pr["vertex_ms"] / pr["substeps"] * 1e3, 2),
                     "substep_alg_bytes_per_tet": round(b_alg, 1)},
    }
    if not args.no_cpu_baseline:
        body.close()
        from oracle import OracleNH
        nh = OracleNH(verts, tets, PP)
        nh.simulate(DT, PP)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 10.0:
            nh.simulate(DT, PP)
            n += 1
        out["cpu_baseline"] = {"value": round(n * len(tets) / (time.perf_counter() - t0) / 1e6, 3), "unit": "M tet-solves/s", "cores": 1,
                               "kind": "port", "sample": "%d substeps of the same lattice, sequential Gauss-Seidel in the caller's tet order "
                                                         "(oracle/tetsim_oracle.c section A: Softbody.js's algorithm, bit-exact with its goldens)" % n}
    return out, body


def beyond_mall(args, device, copy_peak, cells=110, frames=10):
    """SURVEY.md 8(d) asks for a figure at a size beyond the 256 MB Infinity Cache as well: the 110^3-cell lattice (7,986,000 tets,
    ~1.3 GB of per-tet state) on this one GPU, same kernels, `frames` frames after 2 warm-up frames, then the dominant kernel's own
    events over 20 substeps.  Outside the timed region of the headline; ~3 s incl. building the body."""
    from tetsim_amd import SoftBodyHIP, make_lattice
    t_build = time.perf_counter()
    v, t = make_lattice(cells)
    kw = {"constant_rest_shape": True} if args.constant_rest_shape else {}
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", device=device, **kw)
    t_build = time.perf_counter() - t_build
    for _ in range(2):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    el = time.perf_counter() - t0
    finite = bool(np.isfinite(body.pos).all())
    pr = body.profile(SUBSTEPS, DT, PP)
    body.close()
    value = len(t) * SUBSTEPS * frames / el / 1e6
    tet_bytes = TET_KERNEL_BYTES - (48.0 if args.constant_rest_shape else 0.0)
    b_alg = tet_bytes + VERTEX_BYTES * len(v) / len(t)
    tet_us = pr["tet_ms"] / pr["tet_launches"] * 1e3
    ach = tet_bytes * pr["tets_per_tet_launch"] / (tet_us * 1e-6) / 1e9
    res = {"workload": "Kuhn-6 cube lattice %d^3 cells (%d tets, %d particles), same solver and kernels, %d frames of %d substeps" % (cells, len(t), len(v), frames, SUBSTEPS),
           "value": round(value, 1), "unit": "M tet-solves/s", "ms_per_step": round(el / frames * 1e3, 4), "finite": finite,
           "kernel_us": round(tet_us, 2), "vertex_kernel_us": round(pr["vertex_ms"] / pr["vertex_launches"] * 1e3, 2) if pr["vertex_launches"] else 0.0,
           "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
           "substep_achieved": round(b_alg * value * 1e6 / 1e9, 1), "substep_frac": round(b_alg * value * 1e6 / 1e9 / HBM_PEAK_GBS, 4),
           "build_s": round(t_build, 2)}
    if copy_peak.get("1GiB"):
        res["frac_of_1GiB_copy"] = round(ach / copy_peak["1GiB"], 4)
        res["substep_frac_of_1GiB_copy"] = round(b_alg * value * 1e6 / 1e9 / copy_peak["1GiB"], 4)
    return res


def slab_owner(nverts, cells, nz, world):
    """vertex -> rank: whole z-planes, ceil(nz / world) cell layers per slab."""
    plane = (cells + 1) * (cells + 1)
    layers = -(-nz // world)
    return np.minimum((np.arange(nverts) // plane) // layers, world - 1).astype(np.int32)


def make_body(args, cells, scaling, rank, world, local_rank, ranks, vote=False, halo=None):
    """This rank's body of the cells^2 x (cells [x world when weak]) lattice (+ communicator when ranks is not None).
    vote=True: creation (local, may fail on one rank alone: memory, ...) is followed by a vote of all ranks BEFORE the collective
    communicator set-up; if any rank failed, every rank returns (None, ..., error text) instead of hanging in the broadcast."""
    from tetsim_amd import SoftBodyHIP, make_lattice
    nz = cells * world if scaling == "weak" else cells
    pp = dict(PP)
    body, verts, tets, err = None, None, None, None
    try:
        verts, tets = make_lattice(cells, nz=nz)
        kw = {}
        if ranks is not None:
            # the stacked lattice is `world` metres long in z: the reference's hard-coded +-2.5 m clamp (SoftbodyGPU.js:347)
            # would squash it, so N > 1 runs honour physicsParams.worldBounds, widened along z
            zext = 0.5 * (nz / cells) + 2.0
            pp["worldBounds"] = [-2.5, -1.0, -zext, 2.5, 10.0, zext]
            kw = dict(part_count=world, part_index=rank, vert_owner=slab_owner(len(verts), cells, nz, world), ref_fixed_bounds=False)
        if args.constant_rest_shape:
            kw["constant_rest_shape"] = True
        if (halo or args.halo) == "deep" and ranks is not None and world > 1:
            kw["deep_ghosts"] = True
        body = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision=args.precision, device=local_rank, **kw)
    except Exception as e:  # noqa: BLE001
        if not vote:
            raise
        err = "rank %d: %r" % (rank, e)
    if vote and ranks is not None and ranks.min_float(0.0 if err else 1.0) < 1.0:
        if body is not None:
            body.close()
        return None, verts, tets, pp, nz, err or "another rank failed to create its partition"
    if ranks is not None:
        from tetsim_amd import comm_init, comm_unique_id
        uid = ranks.broadcast_bytes(comm_unique_id() if rank == 0 else None, 128)
        comm_init(body, uid, rank, world)
        if (halo or args.halo) in ("p2p", "deep") and world > 1:
            # the peer-to-peer halo on top of the communicator (RCCL keeps carrying the refresh after a dt change): every rank
            # describes its buffers, torch gathers the descriptions, every rank opens its neighbours' (HIP IPC); a local failure is
            # voted on so that no rank steps alone
            from tetsim_amd import p2p_connect, p2p_export
            perr = None
            try:
                blob = p2p_export(body)
            except Exception as e:  # noqa: BLE001
                blob, perr = b"\0" * 512, "rank %d: %r" % (rank, e)
            blobs = ranks.all_gather_bytes(blob, 512)
            if perr is None:
                try:
                    p2p_connect(body, blobs)
                except Exception as e:  # noqa: BLE001
                    perr = "rank %d: %r" % (rank, e)
            if ranks.min_float(0.0 if perr else 1.0) < 1.0:
                if not vote:
                    raise SystemExit("peer-to-peer halo: " + (perr or "another rank could not connect"))
                body.close()
                return None, verts, tets, pp, nz, perr or "another rank could not connect its peer-to-peer halo"
            ranks.barrier()
    return body, verts, tets, pp, nz, None


def timed_frames(body, pp, steps, warmup, ranks):
    """W untimed + K timed frames bracketed by sync + barrier.  Returns (wall seconds of this rank, host seconds this rank spent
    inside the K stepping calls -- the enqueue cost; the calls do not synchronise)."""
    def barrier():
        body.sync()
        if ranks is not None:
            ranks.barrier()

    for _ in range(warmup):
        body.simulateSubsteps(SUBSTEPS, DT, pp)
    barrier()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        body.simulateSubsteps(SUBSTEPS, DT, pp)
        host += time.perf_counter() - h0
    barrier()
    return time.perf_counter() - t0, host


# How an N-rank headline run may be repeated when a transport fails on the node it meets: each rung rebuilds every rank's body with
# more conservative halo settings (the library reads them per body).  A rung is left only by a VOTE of all ranks, and every rank
# runs the same collectives whether its local steps worked or not.
HALO_LADDER = [({}, "flag-synchronised two-queue halo, both chains replayed from captured graphs (the default)"),
               ({"TETSIM_HALO_GRAPH": "0"}, "the same halo path enqueued eagerly (no graph replay)"),
               ({"TETSIM_HALO_SYNC": "events", "TETSIM_HALO_GRAPH": "0"}, "event-synchronised halo path, eager (round 1's)")]


def headline_with_retries(args, cells, rank, world, local_rank, ranks):
    """The timed region of an N-rank run (timed_frames' protocol: W untimed + K timed frames bracketed by synchronise + barrier), with
    every local step caught and voted on; on a failure anywhere all ranks close their bodies and climb one rung of HALO_LADDER.
    Returns (body, verts, tets, pp, nz, wall seconds of this rank, host seconds inside the K calls, [attempt records])."""
    attempts = []
    keys = sorted({k for env, _ in HALO_LADDER for k in env} | {"TETSIM_HALO_TIMEOUT_MS"})
    saved = {k: os.environ.get(k) for k in keys}
    try:
        for rung, (env, what) in enumerate(HALO_LADDER):
            for k in keys:
                if k != "TETSIM_HALO_TIMEOUT_MS":
                    os.environ.pop(k, None) if saved[k] is None else os.environ.__setitem__(k, saved[k])
            os.environ.update(env)
            if saved["TETSIM_HALO_TIMEOUT_MS"] is None:
                os.environ["TETSIM_HALO_TIMEOUT_MS"] = "10000"   # a rank that waits in vain says so after 10 s, not 30
            state = {"err": None}

            def local(fn):
                if state["err"] is None:
                    try:
                        return fn()
                    except Exception as e:  # noqa: BLE001
                        state["err"] = "rank %d: %r" % (rank, e)
                return None

            body, verts, tets, pp, nz, err = make_body(args, cells, args.scaling, rank, world, local_rank, ranks, vote=True)
            el = host = 0.0
            if body is None:
                state["err"] = err
            else:
                for _ in range(args.warmup):
                    local(lambda: body.simulateSubsteps(SUBSTEPS, DT, pp))
                local(body.sync)
                ranks.barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    h0 = time.perf_counter()
                    local(lambda: body.simulateSubsteps(SUBSTEPS, DT, pp))
                    host += time.perf_counter() - h0
                local(body.sync)
                ranks.barrier()
                el = time.perf_counter() - t0
                fin = local(lambda: bool(np.isfinite(body.pos).all()))
                if fin is False:
                    state["err"] = "rank %d: non-finite positions after the timed region" % rank
                if rung == 0 and os.environ.get("TETSIM_BENCH_TEST_FAIL_FIRST_RUNG") == str(rank) and not state["err"]:
                    state["err"] = "rank %d: injected failure (test of the retry ladder)" % rank
            ok = ranks.min_float(0.0 if state["err"] else 1.0) >= 1.0
            attempts.append({"halo": what, "ok": ok} if ok or not state["err"] else {"halo": what, "ok": False, "error_rank%d" % rank: state["err"][:300]})
            if ok:
                return body, verts, tets, pp, nz, el, host, attempts
            if rank == 0:
                print("[bench] N-rank run failed with: %s -- %s" % (what, state["err"] or "an error on another rank"), file=sys.stderr)
            if body is not None:
                try:
                    body.close()
                except Exception:  # noqa: BLE001
                    pass
        raise SystemExit("the N-rank run failed with every halo setting: " + json.dumps(attempts))
    finally:
        for k in keys:
            os.environ.pop(k, None) if saved[k] is None else os.environ.__setitem__(k, saved[k])


def multi_gpu_report(body, world, elapsed_local, host_local, steps, ranks):
    """What makes an N-rank run self-diagnosing: RCCL's own rank count (must equal --gpus), the spread of the ranks' step times,
    this rank's halo volume, the host enqueue time per substep."""
    from tetsim_amd import comm_info
    ci = comm_info(body)
    if ci["rccl_ranks"] != world:
        raise SystemExit("RCCL reports %d ranks in the halo communicator but --gpus is %d: refusing to report a number" % (ci["rccl_ranks"], world))
    ms = elapsed_local / steps * 1e3
    hq = host_local / (steps * SUBSTEPS) * 1e6
    rep = {"rccl_ranks": ci["rccl_ranks"], "halo": ("p2p" if ci.get("p2p") else "rccl") + (" (two-layer ghost region, ghosts every other substep)" if body.info.flags & 32 else ""),
           "ranks_ms_per_step": {"min": round(ranks.min_float(ms), 4), "max": round(ranks.max_float(ms), 4)},
           "host_enqueue_us_per_substep": {"min": round(ranks.min_float(hq), 2), "max": round(ranks.max_float(hq), 2)},
           "halo_rank0": {"neighbours": ci["neighbours"], "send_bytes_per_substep": ci["send_bytes_per_substep"],
                          "recv_bytes_per_substep": ci["recv_bytes_per_substep"], "max_message_bytes": ci["max_message_bytes"]},
           "halo_max_message_bytes_over_ranks": int(ranks.max_float(float(ci["max_message_bytes"]))),
           "owned_tets_rank0": int(body.info.owned_elems), "local_tets_rank0": int(body.info.local_elems)}
    if ci["loopback"]:
        rep["loopback"] = True
    return rep


def p2p_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global):
    """The headline run once more on a fresh body whose halo goes peer to peer (include/tetsim.h: tetsim_halo_p2p_connect): the same
    warm-up and timed frames from the same rest state, so the owned positions must equal the RCCL run's BIT FOR BIT -- on real
    peers, which the one-GPU tests cannot show -- and the rate says what taking RCCL's send/recv kernel off the substep's chain
    is worth here.  Every local step is caught and VOTED on (a rank never leaves the others inside a collective), device-side
    waits are short, one probe substep comes first, and the whole leg sits under the HeadlineGuard's budget."""
    saved = os.environ.get("TETSIM_HALO_TIMEOUT_MS")
    os.environ["TETSIM_HALO_TIMEOUT_MS"] = "4000"
    state = {"err": None}

    def local(fn):      # run a local step unless this rank has failed already; remember the first failure
        if state["err"] is None:
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                state["err"] = "rank %d: %r" % (rank, e)
        return None

    def everyone_ok():  # collective
        return ranks.min_float(0.0 if state["err"] else 1.0) >= 1.0

    body2 = None
    try:
        body2, _, _, pp2, _, err = make_body(args, cells, args.scaling, rank, world, local_rank, ranks, vote=True, halo="p2p")
        if body2 is None:
            return {"error": err}
        local(lambda: (body2.simulate(DT, pp2), body2.sync()))   # a transport that does not work shows here, within seconds
        if not everyone_ok():
            return {"error": state["err"] or "the probe substep failed on another rank"}
        local(lambda: body2.simulateSubsteps(SUBSTEPS - 1, DT, pp2))
        # the first frame is done; the others as in the headline run: the rest of the warm-up untimed, then the timed frames
        frames_before = max(args.warmup - 1, 0)
        timed = args.warmup + args.steps - 1 - frames_before
        for _ in range(frames_before):
            local(lambda: body2.simulateSubsteps(SUBSTEPS, DT, pp2))
        local(body2.sync)
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(timed):
            local(lambda: body2.simulateSubsteps(SUBSTEPS, DT, pp2))
        local(body2.sync)
        ranks.barrier()
        el_local = time.perf_counter() - t0
        pos2 = local(lambda: body2.pos)
        same = pos2 is not None and pos_rccl is not None and bool(np.array_equal(pos2.view(np.uint32), pos_rccl.view(np.uint32)))
        fin = pos2 is not None and bool(np.isfinite(pos2).all())
        el = ranks.max_float(el_local)
        res = {"value": round(nt_global * SUBSTEPS * timed / el / 1e6, 1) if timed > 0 else None, "unit": "M tet-solves/s",
               "ms_per_step": round(el / max(timed, 1) * 1e3, 4), "steps": timed,
               "bit_equal_to_rccl_run": bool(ranks.min_float(1.0 if same else 0.0) >= 1.0), "finite": bool(ranks.min_float(1.0 if fin else 0.0) >= 1.0),
               "ranks_ms_per_step": {"min": round(ranks.min_float(el_local / max(timed, 1) * 1e3), 4), "max": round(ranks.max_float(el_local / max(timed, 1) * 1e3), 4)}}
        if not everyone_ok():
            res["error"] = state["err"] or "a step failed on another rank"
        return res
    finally:
        if body2 is not None:
            try:
                ranks.barrier()
                body2.close()
            except Exception:  # noqa: BLE001
                pass
        if saved is None:
            os.environ.pop("TETSIM_HALO_TIMEOUT_MS", None)
        else:
            os.environ["TETSIM_HALO_TIMEOUT_MS"] = saved


def promote_p2p(out, res, steps, world, mode="best"):
    """The headline of an N-rank run is the faster of the two halo transports -- if the peer-to-peer run (`res`, p2p_check) is VALIDATED
    in this very run: the same frames from the same rest state under the same protocol (barrier, synchronise, max over ranks), every
    rank's positions equal to the RCCL run's bit for bit.  Otherwise, or with mode "rccl", the RCCL figures in `out` stand.  Returns
    whether `out` was changed (value, ms_per_step, multi_gpu.halo / ranks_ms_per_step / rccl_halo, config.parallelism)."""
    if mode != "best" or not isinstance(res, dict) or res.get("error") or not res.get("bit_equal_to_rccl_run") or not res.get("finite"):
        return False
    if res.get("steps") != steps or not res.get("value") or res["value"] <= out["value"]:
        return False
    mgr = out["multi_gpu"]
    mgr["rccl_halo"] = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "ranks_ms_per_step": mgr.get("ranks_ms_per_step")}
    out["value"], out["ms_per_step"] = res["value"], res["ms_per_step"]
    mgr["ranks_ms_per_step"] = res.get("ranks_ms_per_step")
    mgr["halo"] = ("p2p: boundary particles stored straight into the neighbours' IPC-mapped ghost ranges -- the faster of the two transports, validated in "
                   "this run (positions bit-equal to the RCCL run of the same frames, whose figures are in multi_gpu.rccl_halo)")
    out["config"]["parallelism"] = "z-slab domain decomposition x%d, peer-to-peer ghost halo per substep (RCCL for set-up and validation)" % world
    return True


def pmc_traffic(kname, kernel_sha):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json) -- only if they were taken on
    THIS kernel build (same kernel_sha); a stale figure is reported as null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        return t.get(kname, {}).get("hbm_bytes_per_launch") if t.get("kernel_sha") == kernel_sha else None
    except Exception:
        return None


def run(args, rank, world, local_rank, ranks):
    """One rank of the benchmark.  `ranks` is None (single process, no communicator) or an adapter with broadcast_bytes /
    barrier / max_float / min_float."""
    use_dist = ranks is not None
    from tetsim_amd import library_info, measure_copy_bandwidth

    cells = args.cells
    if args.solver == "neohookean":
        if world > 1:
            raise SystemExit("--solver neohookean is a single-GPU benchmark: Gauss-Seidel would need one halo per colour (replicas only)")
        from tetsim_amd import make_lattice
        verts, tets = make_lattice(cells)
        out, body = run_neohookean(args, verts, tets, local_rank)
        out["library"] = library_info()
        return out, body
    attempts = []
    if use_dist and world > 1:
        body, verts, tets, pp, nz, elapsed_local, host_local, attempts = headline_with_retries(args, cells, rank, world, local_rank, ranks)
    else:
        body, verts, tets, pp, nz, _ = make_body(args, cells, args.scaling, rank, world, local_rank, ranks)
        # ---- timed region --------------------------------------------------------------------------------
        elapsed_local, host_local = timed_frames(body, pp, args.steps, args.warmup, ranks)
        if not np.isfinite(body.pos).all():
            raise SystemExit("non-finite positions after the timed region")
    nt_global = len(tets)
    elapsed = ranks.max_float(elapsed_local) if use_dist else elapsed_local
    mg = multi_gpu_report(body, world, elapsed_local, host_local, args.steps, ranks) if use_dist else None
    if mg is not None and len(attempts) > 1:
        mg["halo_attempts"] = attempts   # (the ones before the last failed: the headline was measured with the last one's settings)

    lib = library_info()
    out = None
    if rank == 0:
        value = nt_global * SUBSTEPS * args.steps / elapsed / 1e6
        nv_global = len(verts)
        out = {
            "metric": "tet_solves_per_sec", "value": round(value, 1), "unit": "M tet-solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Kuhn-6 cube lattice %dx%dx%d cells (%d tets, %d particles), polar-decomposition Jacobi, "
                                   "%d substeps/frame, dt=1/1200 s" % (cells, cells, nz, nt_global, nv_global, SUBSTEPS),
                       "solver": "polar_jacobi", "arithmetic": args.precision,
                       "formulation": "constant rest shape (opt-in)" if args.constant_rest_shape else "reference (rest shape carried from substep to substep, 148 B/tet)", "substeps_per_step": SUBSTEPS,
                       "rotation_exit": ("iteration 1: |omega| < 1e-9 (the reference's, SoftbodyGPU.js:131); correction iterations 2..9: |omega| < 1e-6 rad "
                                         "(FAST default; other_configs.config3_reference_threshold has the same frames with 1e-9 throughout)") if args.precision == "fast"
                                        else "|omega| < 1e-9 (the reference's, SoftbodyGPU.js:131)",
                       "tets": nt_global, "particles": nv_global,
                       "parallelism": "single GPU" if world == 1 else "z-slab domain decomposition x%d, RCCL ghost halo per substep" % world},
            "library": lib,
        }
        if world == 1:
            out["host_enqueue_us_per_substep"] = round(host_local / (args.steps * SUBSTEPS) * 1e6, 2)
        if mg is not None:
            out["multi_gpu"] = mg
    # dominant kernel: its OWN begin/end HIP events (hipExtLaunchKernelGGL) on the handle's stream, inside the real
    # tet -> particle -> tet ... sequence, 60 substeps right after the timed region (same kernels as the graph).  N > 1: every
    # rank takes part (the substeps exchange halos as usual); rank 0 reports ITS interior tet kernel -- the boundary tiles run
    # beside it on the halo stream.
    pr = None
    if world == 1 or (args.profile_ranks and args.precision == "fast"):
        # three batches of 60 substeps, the median batch is reported (a single batch right after the timed region is
        # occasionally 5-8% slow on a box that agrees with rocprofv3 otherwise)
        batches = sorted((body.profile(SUBSTEPS * 3, DT, pp) for _ in range(3)), key=lambda p: p["tet_ms"] / p["tet_launches"])
        pr = batches[1]
        body.sync()
        if use_dist:
            ranks.barrier()
    tet_bytes = TET_KERNEL_BYTES - (48.0 if args.constant_rest_shape else 0.0)   # constant rest shape: read only, never written back
    b_alg = tet_bytes + VERTEX_BYTES * len(verts) / len(tets)
    if rank == 0 and pr is not None:
        tet_us = pr["tet_ms"] / pr["tet_launches"] * 1e3
        vert_us = pr["vertex_ms"] / pr["vertex_launches"] * 1e3 if pr["vertex_launches"] else 0.0
        units = pr["tets_per_tet_launch"]
        kname = "pjb_tet_kernel" if args.precision == "fast" else "pj_tet_kernel_precise"
        if body.info.fused_particle_pass:   # small bodies (< 2,048 tiles): one kernel per substep does the particle row too
            kname, tet_bytes = "pjb_tet_fused_kernel", b_alg
        achieved = tet_bytes * units / (tet_us * 1e-6) / 1e9
        traffic = pmc_traffic(kname, lib["kernel_sha"]) if world == 1 and not args.constant_rest_shape and cells == CELLS else None
        out["roofline"] = {"bound": "hbm", "kernel": kname + ("" if world == 1 else " (rank 0, interior tiles)"),
                           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                           "kernel_us": round(tet_us, 2), "vertex_kernel_us": round(vert_us, 2),
                           "alg_bytes_per_launch": tet_bytes * units,
                           "substep_alg_bytes_per_tet": round(b_alg, 1),
                           "substep_achieved": round(b_alg * out["value"] * 1e6 / 1e9, 1),
                           "substep_frac": round(b_alg * out["value"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)}
        if world == 1:
            # SURVEY.md 8(d) "bounding roofline": the peak is also MEASURED on this box -- a device copy at the footprint class of
            # the 1 M-tet working set (fits the 256 MB Infinity Cache) and at 1 GiB (streams from HBM)
            # (the tuned probe of tetsim_measure_stream_bandwidth: four independent 16-byte accesses per lane, plain / non-temporal and the
            # grid size chosen at first use; read-only and write-only rates beside the copy rate)
            from tetsim_amd import measure_stream_bandwidth
            cp = {"64MiB": round(measure_copy_bandwidth(64 << 20, 20), 0), "1GiB": round(measure_copy_bandwidth(1 << 30, 10), 0)}
            out["roofline"]["measured_copy_peak"] = cp
            out["roofline"]["measured_stream_peak_1GiB"] = {"read": round(measure_stream_bandwidth(1 << 30, "read", 10), 0),
                                                            "write": round(measure_stream_bandwidth(1 << 30, "write", 10), 0), "unit": "GB/s"}
            out["roofline"]["frac_of_measured_peak"] = {"kernel_vs_64MiB_copy": round(achieved / cp["64MiB"], 4),
                                                        "kernel_vs_1GiB_copy": round(achieved / cp["1GiB"], 4),
                                                        "substep_vs_64MiB_copy": round(b_alg * out["value"] * 1e6 / 1e9 / cp["64MiB"], 4),
                                                        "substep_vs_1GiB_copy": round(b_alg * out["value"] * 1e6 / 1e9 / cp["1GiB"], 4)}
    def whole_job_roofline():
        # N > 1 without --profile-ranks: the whole-job figure only (no extra GPU work after the timed region)
        agg = b_alg * out["value"] * 1e6 / 1e9
        return {"bound": "hbm", "kernel": "whole substep, all ranks (tet + particle kernels)", "achieved": round(agg, 1),
                "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(agg / (HBM_PEAK_GBS * world), 4), "traffic": None,
                "substep_alg_bytes_per_tet": round(b_alg, 1)}
    if rank == 0 and pr is None:
        out["roofline"] = whole_job_roofline()
    # ---- optional legs of an N-rank run: nothing below may cost the headline (HeadlineGuard) ------------------------------------
    if use_dist and world > 1:
        GUARD.arm(out, int(os.environ.get("TETSIM_BENCH_OPTIONAL_S", "240")), "the legs after the headline (peer-to-peer halo check / config 5)")
    if use_dist and world > 1 and args.halo == "rccl" and (args.p2p_check == "on" or (args.p2p_check == "auto" and not args.fake_ranks)):
        try:
            pos_rccl = body.pos
        except Exception:  # noqa: BLE001
            pos_rccl = None
        res = p2p_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global)
        if rank == 0:
            out["multi_gpu"]["p2p_halo"] = res
            if promote_p2p(out, res, args.steps, world, args.headline_halo):
                if pr is None:
                    out["roofline"] = whole_job_roofline()
                else:
                    out["roofline"]["substep_achieved"] = round(b_alg * out["value"] * 1e6 / 1e9, 1)
                    out["roofline"]["substep_frac"] = round(b_alg * out["value"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)
    # ---- BASELINE config 5, literally: the 110^3-cell lattice (7,986,000 tets) cut into N slabs -- strong scaling -----------
    if use_dist and (args.config5 == "on" or (args.config5 == "auto" and world == 8)) and not (args.scaling == "strong" and cells == args.config5_cells):
        # (the headline body stays alive: if this second body cannot be built on some rank, the line above is still reported)
        body5, v5, t5, pp5, _, err5 = make_body(args, args.config5_cells, "strong", rank, world, local_rank, ranks, vote=True)
        if body5 is None:
            if rank == 0:
                out["config5_strong"] = {"error": err5}
        else:
            body.close()
            body = body5
            steps5 = max(1, min(args.steps, 10))
            e5_local, h5_local = timed_frames(body, pp5, steps5, min(args.warmup, 2), ranks)
            e5 = ranks.max_float(e5_local)
            finite = ranks.min_float(1.0 if np.isfinite(body.pos).all() else 0.0)
            mg5 = multi_gpu_report(body, world, e5_local, h5_local, steps5, ranks)
            if rank == 0:
                v = len(t5) * SUBSTEPS * steps5 / e5 / 1e6
                out["config5_strong"] = {
                    "workload": "Kuhn-6 cube lattice %d^3 cells (%d tets, %d particles) cut into %d z-slabs, %d-particle interface planes, "
                                "polar-decomposition Jacobi, %d substeps/frame" % (args.config5_cells, len(t5), len(v5), world, (args.config5_cells + 1) ** 2, SUBSTEPS),
                    "scaling": "strong", "value": round(v, 1), "unit": "M tet-solves/s", "steps": steps5, "ms_per_step": round(e5 / steps5 * 1e3, 4),
                    "finite": bool(finite), "multi_gpu": mg5,
                    "substep_frac_of_hbm_roofline": round(b_alg * v * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)}
    GUARD.disarm()
    if world == 1:
        body.close()
        if not args.no_beyond_mall and args.precision == "fast" and cells == CELLS and args.solver == "polar" and "roofline" in out:
            out["roofline"]["beyond_mall"] = beyond_mall(args, local_rank, out["roofline"].get("measured_copy_peak", {}))
        if not args.no_other_configs and args.precision == "fast" and cells == CELLS:
            out["other_configs"] = other_configs(args.steps, args.warmup)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(verts, tets)
    return out, body


if __name__ == "__main__":
    main()
