"""bench.py as its own launcher (CPU): `python bench.py --gpus N` with no WORLD_SIZE in the environment starts N rank processes,
forwards rank 0's one JSON line, propagates a failing rank's exit code and stops the survivors.  The ranks here are a stub
(tests/stubs/rank_stub.py); the real ranks are exercised on the GPU box by tests/test_bench_gpu.py."""
import importlib.util
import io
import json
import os
import subprocess
import sys
import time

from conftest import ROOT

STUB = [sys.executable, os.path.join(ROOT, "tests", "stubs", "rank_stub.py")]


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _launch(n, env=None, **kw):
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        out, err = io.StringIO(), io.StringIO()
        t0 = time.monotonic()
        code = _bench().self_launch(n, ["--gpus", str(n), "--steps", "2"], worker=STUB, devices=kw.pop("devices", n), out=out, err=err, **kw)
        return code, out.getvalue(), err.getvalue(), time.monotonic() - t0
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_rank_zero_line_is_forwarded_and_nothing_else():
    code, out, err, _ = _launch(4)
    assert code == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["rank"] == 0 and d["world"] == 4 and d["local_rank"] == 0 and d["addr"] == "127.0.0.1" and 1024 < d["port"] < 65536
    assert d["argv"] == ["--gpus", "4", "--steps", "2"]
    assert err.count("noise from rank") == 3          # the other ranks' stdout goes to stderr


def test_failing_rank_takes_the_launch_down_with_its_exit_code():
    code, out, err, took = _launch(3, env={"STUB_FAIL_RANK": "1", "STUB_FAIL_CODE": "7", "STUB_HANG": "1"})
    assert code == 7 and out.strip() == "" and "rank 1 exited with 7" in err
    assert took < 20                                   # the hanging ranks were stopped, not waited for


def test_too_few_devices_is_a_clear_message_not_a_usage_hint():
    code, out, err, _ = _launch(8, devices=0)
    assert code == 2 and out == "" and "8 devices requested, 0 visible" in err


def test_hung_ranks_hit_the_limit():
    code, _, err, took = _launch(2, env={"STUB_HANG": "1"}, limit_s=1.0)
    assert code == 124 and "still running" in err and took < 20


def test_plain_python_launch_on_a_box_without_gpus():
    """The driver's N = 1 command with N changed, on this GPU-less container: fails fast, names the reason."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.device_count() >= 8:
        return
    assert r.returncode == 2 and r.stdout == "" and "8 devices requested, %d visible" % torch.cuda.device_count() in r.stderr


def test_headline_guard_prints_the_headline_when_an_optional_leg_hangs():
    """bench.py's optional N > 1 legs (peer-to-peer halo check, config 5) run under a budget: when it runs out, rank 0 prints the
    headline line it already has -- on the REAL stdout, also while fd 1 is routed to stderr -- with a note, and the process leaves
    with exit code 0; the other ranks leave silently."""
    code = (
        "import sys, time; sys.path.insert(0, %r)\n"
        "import importlib.util\n"
        "spec = importlib.util.spec_from_file_location('bench_module', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "line = {'metric': 'tet_solves_per_sec', 'value': 1.0} if sys.argv[1] == '0' else None\n"
        "with b.stdout_to_stderr():\n"
        "    b.GUARD.arm(line, 1, 'the test leg')\n"
        "    print('noise that must stay off the real stdout')\n"
        "    time.sleep(30)\n"
        "print('never reached')\n" % (ROOT, os.path.join(ROOT, "bench.py")))
    for rank, want_line in (("0", True), ("1", False)):
        t0 = time.monotonic()
        r = subprocess.run([sys.executable, "-c", code, rank], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and time.monotonic() - t0 < 20
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        if want_line:
            assert len(lines) == 1
            d = json.loads(lines[0])
            assert d["value"] == 1.0 and "cut short" in d["notes"][0]
        else:
            assert lines == []
    # and a leg that finishes in time leaves the line alone
    code2 = code.replace("time.sleep(30)", "b.GUARD.disarm(); time.sleep(1.5)").replace("print('never reached')", "print('reached')")
    r = subprocess.run([sys.executable, "-c", code2, "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "reached"


def test_the_headline_of_an_n_rank_run_is_the_faster_transport_only_when_validated():
    """bench.py promote_p2p: the peer-to-peer halo's figures replace the RCCL run's on the headline line only if that run was validated
    (bit-equal positions, finite, the same number of timed frames, no error) AND faster; the RCCL figures stay beside them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module_promote", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)

    def line():
        return {"value": 100.0, "unit": "M tet-solves/s", "ms_per_step": 8.0, "config": {"parallelism": "z-slab domain decomposition x8, RCCL ghost halo per substep"},
                "multi_gpu": {"halo": "rccl", "ranks_ms_per_step": {"min": 7.9, "max": 8.0}}}
    good = {"value": 125.0, "unit": "M tet-solves/s", "ms_per_step": 6.4, "steps": 20, "bit_equal_to_rccl_run": True, "finite": True,
            "ranks_ms_per_step": {"min": 6.3, "max": 6.4}}
    out = line()
    assert b.promote_p2p(out, dict(good), 20, 8) is True
    assert out["value"] == 125.0 and out["ms_per_step"] == 6.4 and out["multi_gpu"]["halo"].startswith("p2p") and "peer-to-peer" in out["config"]["parallelism"]
    assert out["multi_gpu"]["rccl_halo"] == {"value": 100.0, "unit": "M tet-solves/s", "ms_per_step": 8.0, "ranks_ms_per_step": {"min": 7.9, "max": 8.0}}
    assert out["multi_gpu"]["ranks_ms_per_step"] == {"min": 6.3, "max": 6.4}
    for bad in (dict(good, bit_equal_to_rccl_run=False), dict(good, finite=False), dict(good, error="rank 3: TetSimError"), dict(good, steps=19),
                dict(good, value=99.0), dict(good, value=None), {"error": "no peer access"}, None):
        out = line()
        assert b.promote_p2p(out, bad, 20, 8) is False and out == line()
    out = line()
    assert b.promote_p2p(out, dict(good), 20, 8, mode="rccl") is False and out == line()
