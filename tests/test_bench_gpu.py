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
            "dtype", "data", "config", "roofline"}


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


def test_neohookean_line_on_request():
    """`--solver neohookean` (BASELINE config 4): same metric and JSON shape, never the default."""
    d = _run(["--solver", "neohookean", "--order", "clustered", "--cells", "10", "--steps", "3", "--warmup", "1", "--precision", "precise", "--no-cpu-baseline"])
    assert REQUIRED <= set(d) and d["config"]["solver"] == "neohookean_gs" and d["dtype"] == "f64" and d["value"] > 0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["substep_alg_bytes_per_tet"] > 56


@pytest.mark.parametrize("extra", [[], ["--profile-ranks"], ["--scaling", "strong"]])
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
