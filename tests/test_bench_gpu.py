"""bench.py's contract on the GPU box: exactly one JSON line on stdout with the fields the driver reads, for the single-GPU
run and for the multi-rank code path (ranks = host threads on this one GPU, librccl = tests/mock_rccl)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "library"}


def _run(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE line, nothing else on stdout
    return json.loads(lines[0])


def test_single_gpu_line():
    d = _run(["--cells", "14", "--steps", "3", "--warmup", "1"])
    assert REQUIRED | {"cpu_baseline"} <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["metric"] == "tet_solves_per_sec" and d["unit"] == "M tet-solves/s" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert abs(d["value"] - d["config"]["tets"] * 20 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    lib = d["library"]
    assert lib["abi"] == 5 and lib["ablation"] is False and lib["debug_env"] == [] and len(lib["kernel_sha"]) == 16
    assert rf["traffic"] is None          # not the headline lattice: no counter figure is attached to it
    # the dominant kernel is timed over a replay of the timed frames, which must retrace them bit for bit; the window after them beside it
    assert "bit-equal to the timed one: True" in rf["fast_exit"]["window"] and rf["after_timed_region"]["kernel_us"] > 0
    # ... and the fraction the line LEADS with is the equal-work one: nine rotation iterations per tet (the reference's threshold over the
    # timed frames), the FAST-exit kernel the value ran and the on-floor window beside it, each with its own achieved / frac
    # (round 6: `frac` itself is that kernel INSIDE the graphs tetsim_step_n replays -- the timed frames' wall clock with the reference's
    # threshold minus the particle kernel and the launch boundaries --, the per-launch event figure on the floor beside it as frac_events)
    rt = rf["timed_frames_reference_threshold"]
    assert "AS GRAPH REPLAYS" in rf["window"] and "nine rotation iterations" in rf["work"] and rf["frac"] == rt["in_graph"]["frac_implied"]
    assert rf["frac_events"] == rf["on_floor"]["frac"] and "lies on the floor" in rf["window_events"] and rf["kernel_us"] == rt["in_graph"]["kernel_us_implied"]
    assert rt["kernel_us"] >= 0.97 * rf["fast_exit"]["kernel_us"] and d["value_reference_threshold"] > 0
    assert len(d["value_reference_threshold_runs"]) == 3 and 0 < rf["frac_substep_reference_threshold"] < 1
    # the lean tet record beside the reference formulation: its own value, its own B_alg, its own roofline object
    rl = d["roofline_lean"]
    assert d["value_lean"] > 0 and d["value_lean_reference_threshold"] > 0 and len(d["value_lean_runs"]) == 3 and rl["alg_bytes_per_tet"] == 92.0
    assert rl["kernel"] == "pjb_tet_kernel_lean" and rl["finite"] is True and abs(rl["frac"] - rl["achieved"] / 8000.0) < 1e-3 and rl["on_floor"]["kernel_us"] > 0


def test_headline_line_carries_every_baseline_config():
    """The default workload (1 M-tet lattice): `other_configs` reports BASELINE configs 1, 2 and 4 next to the headline
    (config 3), and roofline.traffic is either null or keyed to this very kernel build."""
    d = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["config"]["tets"] == 998250
    rf = d["roofline"]
    # round 6: tetsim_step_n of this body is ONE launch per call (pjb_call_kernel: tiles + particles of the 20 substeps) -- THAT is the dominant
    # kernel, timed by its own events around every launch, priced with the whole substep's algorithmic bytes; equal work (reference threshold) leads
    assert "pjb_call_kernel<0>" in rf["kernel"] and "reference's rotation threshold" in rf["window"] and "tetsim_time_step_n" in rf["window"] and rf["launches"] == 3
    assert abs(rf["alg_bytes_per_launch"] - rf["alg_bytes_per_tet_solve"] * 998250 * 20) / rf["alg_bytes_per_launch"] < 1e-3 and 170 < rf["alg_bytes_per_tet_solve"] < 176
    assert abs(rf["frac"] - rf["alg_bytes_per_launch"] / (rf["kernel_us"] * 1e-6) / 1e9 / 8000.0) < 2e-3 and abs(rf["substep_us"] - rf["kernel_us"] / 20) < 0.01
    for k in ("on_floor", "fast_exit", "fast_exit_on_floor"):
        assert rf[k]["kernel_us"] > 0 and 0 < rf[k]["frac"] < 1
    assert rf["fast_exit"]["kernel_us"] <= rf["kernel_us"] * 1.03          # the FAST exit never does more work than the reference's threshold
    assert abs(rf["headline_wall_clock"]["substep_us"] - d["ms_per_step"] * 1e3 / 20) < 0.01
    # what rounds 1-5 measured -- the tet kernel + particle kernel pair, which tetsim_step / tetsim_profile still run, same bits -- beside it
    tk = rf["two_kernel_path"]
    assert "bit-equal to the timed one: True" in tk["fast_exit"]["window"] and "(60 launches)" in tk["fast_exit"]["window"] and "pjb_tet_kernel" in tk["kernel"]
    rt = tk["timed_frames_reference_threshold"]
    assert "(60 launches)" in rt["window"] and "in_graph" not in rt and tk["on_floor"]["kernel_us"] > 0 and tk["frac"] == tk["on_floor"]["frac"]
    assert rt["frac"] <= tk["fast_exit"]["frac"] * 1.03 and 0 < d["value_reference_threshold"] <= d["value"] * 1.05
    # the lean record is faster than the reference formulation, through the same one-launch call
    rl = d["roofline_lean"]
    assert d["value_lean"] > d["value"] and d["value_lean_reference_threshold"] > d["value_reference_threshold"] and "pjb_call_kernel<2>" in rl["kernel"]
    assert rl["substep_us"] < rf["substep_us"] and rl["alg_bytes_per_tet"] == 92.0 and rl["two_kernel_path"]["on_floor"]["kernel_us"] < tk["on_floor"]["kernel_us"]
    # the committed rocprofv3 summary of this command and the kernel's ceiling, keyed to the kernel build they were taken on
    meta = json.load(open(os.path.join(ROOT, "profiles", "bench_kernel_stats.json")))
    if "rocprof" in rf:
        assert rf["rocprof"]["file"] == "profiles/" + meta["csv"] and rf["rocprof"]["stale"] == (meta["kernel_sha"] != d["library"]["kernel_sha"])
        assert abs(rf["frac_rocprof"] - rf["alg_bytes_per_launch"] / (rf["rocprof"]["kernel_us"] * 1e-6) / 1e9 / 8000.0) < 1e-3
    ce = tk["ceiling"]
    assert ce["ceiling_us"] == max(ce["memory_floor_us"], ce["valu_issue_floor_us"]) and 0.5 < ce["kernel_vs_ceiling"] <= 1.0   # (against the event figure: the floors are event-timed)
    oc = d["other_configs"]
    c1, c2, c4 = oc["config1_dragon_neohookean_cpu_path"], oc["config2_dragon_polar_jacobi"], oc["config4_lattice_1m_neohookean_gs_vs_jacobi"]
    assert c1["hip_original_order_precise"]["value"] > 0 and c1["hip_coloured_precise"]["value"] > c1["hip_original_order_precise"]["value"]
    assert c2["fast"]["value"] > 50 and c2["fast"]["us_per_substep"] < 100 and c2["precise"]["value"] > 0
    for k in ("neohookean_clustered_gs_fast", "neohookean_clustered_gs_precise", "polar_jacobi_fast"):
        r = c4[k]["mean_abs_detF_minus_1_after_1_5_30_frames"]
        assert c4[k]["value"] > 1000 and len(r) == 3 and all(0 <= x < 0.5 for x in r)
    # Gauss-Seidel holds the volume under contact where one Jacobi iteration per substep goes soft
    assert c4["neohookean_clustered_gs_precise"]["mean_abs_detF_minus_1_after_1_5_30_frames"][2] < c4["polar_jacobi_fast"]["mean_abs_detF_minus_1_after_1_5_30_frames"][2]
    tr = tk["traffic"]
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert (tr is None) == (pmc.get("kernel_sha") != d["library"]["kernel_sha"])


def test_neohookean_line_on_request():
    """`--solver neohookean` (BASELINE config 4): same metric and JSON shape, never the default."""
    d = _run(["--solver", "neohookean", "--order", "clustered", "--cells", "10", "--steps", "3", "--warmup", "1", "--precision", "precise", "--no-cpu-baseline"])
    assert REQUIRED <= set(d) and d["config"]["solver"] == "neohookean_gs" and d["dtype"] == "f64" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["substep_alg_bytes_per_tet"] > 56


def _mock_rccl():
    here = os.path.join(ROOT, "tests", "mock_rccl")
    lib = os.path.join(here, "libmock_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(os.path.join(here, "mock_rccl.cpp")):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-std=c++17", os.path.join(here, "mock_rccl.cpp"), "-o", lib],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    return lib


def test_a_failed_n_rank_run_is_repeated_with_more_conservative_halo_settings():
    """bench.py's retry ladder (headline_with_retries): a failure on ANY rank is voted on, every rank closes its body, and all of them
    rebuild -- new communicator, new body -- with the halo path enqueued eagerly; the line carries the attempts.  The failure is
    injected on rank 1 after the first rung's timed region (TETSIM_BENCH_TEST_FAIL_FIRST_RUNG)."""
    d = _run(["--fake-ranks", "3", "--cells", "12", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
             env={"TETSIM_RCCL_LIB": _mock_rccl(), "TETSIM_BENCH_TEST_FAIL_FIRST_RUNG": "1"})
    assert REQUIRED <= set(d) and d["n_gpus"] == 3 and d["value"] > 0
    att = d["multi_gpu"]["halo_attempts"]
    # the order of an N-rank run (round 6): the peer-to-peer halo first -- thread-ranks of one process cannot step it independently: recorded
    # as skipped --, then RCCL with the default settings (the injected failure), then RCCL enqueued eagerly
    assert len(att) == 3 and "peer-to-peer" in att[0]["halo"] and "skipped" in att[0] and att[1]["ok"] is False and "default" in att[1]["halo"] and att[2]["ok"] is True and "eagerly" in att[2]["halo"]
    assert d["multi_gpu"]["rccl_ranks"] == 3 and d["multi_gpu"]["halo"].startswith("rccl")
    # --halo rccl: RCCL first, no peer-to-peer rung at all
    d = _run(["--fake-ranks", "3", "--cells", "12", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--halo", "rccl"], env={"TETSIM_RCCL_LIB": _mock_rccl()})
    assert "halo_attempts" not in d["multi_gpu"] and "RCCL ghost halo" in d["config"]["parallelism"]


@pytest.mark.parametrize("extra", [[], ["--profile-ranks"], ["--scaling", "strong"], ["--config5", "on", "--config5-cells", "18"]])
def test_multi_rank_code_path_with_thread_ranks(extra):
    here = os.path.join(ROOT, "tests", "mock_rccl")
    lib = os.path.join(here, "libmock_rccl.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(os.path.join(here, "mock_rccl.cpp")):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O1", "-std=c++17", os.path.join(here, "mock_rccl.cpp"), "-o", lib],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    cells = 24 if "--profile-ranks" in extra else 12      # per-kernel profiling needs interior tiles on every rank
    d = _run(["--fake-ranks", "3", "--cells", str(cells), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra, env={"TETSIM_RCCL_LIB": lib})
    assert REQUIRED <= set(d) and d["n_gpus"] == 3 and d["value"] > 0
    cells_z = cells if "strong" in extra else 3 * cells
    assert d["config"]["tets"] == cells * cells * cells_z * 6 and d["scaling"] == ("strong" if "strong" in extra else "weak")
    assert "x3" in d["config"]["parallelism"]
    if "--profile-ranks" in extra:
        assert "interior tiles" in d["roofline"]["kernel"] and d["roofline"]["kernel_us"] > 0
    else:
        assert d["roofline"]["peak"] == 3 * 8000.0
    # what makes a real N-rank run self-diagnosing: RCCL's own rank count, the spread over ranks, halo volume, host enqueue time
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == 3 and 0 < mg["ranks_ms_per_step"]["min"] <= mg["ranks_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001
    assert 0 < mg["host_enqueue_us_per_substep"]["min"] <= mg["host_enqueue_us_per_substep"]["max"]
    plane = (cells + 1) ** 2
    assert mg["halo_rank0"] == {"neighbours": 1, "send_bytes_per_substep": 16 * plane, "recv_bytes_per_substep": 16 * plane, "max_message_bytes": 16 * plane}
    assert mg["halo_max_message_bytes_over_ranks"] == 16 * plane
    # ... and the transfer term of the halo chain, measured on the transport the run used (100 exchanges of the real messages)
    hx = mg["halo_exchange_us"]
    assert 0 < hx["rank0"]["min"] <= hx["rank0"]["median"] <= hx["rank0"]["max"] and hx["median_min_over_ranks"] <= hx["median_max_over_ranks"]
    if "--config5" in extra:
        c5 = d["config5_strong"]
        assert c5["scaling"] == "strong" and c5["value"] > 0 and c5["finite"] is True and "18^3" in c5["workload"] and "3 z-slabs" in c5["workload"]
        assert c5["multi_gpu"]["rccl_ranks"] == 3 and c5["multi_gpu"]["halo_rank0"]["max_message_bytes"] == 16 * 19 * 19
    else:
        assert "config5_strong" not in d


@pytest.mark.parametrize("how", [["--force-dist"], ["--self-spawn"]])
def test_real_torch_distributed_adapter_with_one_rank(how):
    """The five torch.distributed calls of the real launch (TorchRanks: init over RCCL, byte broadcast of the communicator id,
    barrier, max / min over ranks) and libtetsim's own RCCL communicator, with ONE real rank: once in this process
    (--force-dist) and once as a rank process started by bench.py's own launcher (--self-spawn = what `python bench.py --gpus N`
    does for N > 1 when no launcher set WORLD_SIZE)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--cells", "14", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"] + how, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == 1 and mg["halo_rank0"]["neighbours"] == 0


def test_plain_python_launch_asks_for_more_devices_than_there_are():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and r.stdout == "" and "%d devices requested, %d visible" % (n, n - 1) in r.stderr
