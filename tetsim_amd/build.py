"""Builds libtetsim_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m tetsim_amd.build [--force]

Per-file flags matter: the PRECISE translation units and the host preprocessing reproduce the reference's
rounding and must not fuse multiply-add (-ffp-contract=off); the FAST units are built with contraction on.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libtetsim_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# -fno-slp-vectorize: on gfx950 v_pk_{mul,add,fma}_f32 issue at half the rate of their scalar forms (no flop gain)
# and the packing costs v_mov shuffles + VGPRs; the rotation-extraction loop is 109 -> ~100 VALU, 340 -> 200 cycles
# per iteration, and the tet kernels drop from ~104 to ~52 VGPRs without it (rocprof + ISA notes in DESIGN.md).
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-fno-slp-vectorize",
          f"--offload-arch={ARCH}"]
UNITS = {
    "host_prep.cpp": ["-ffp-contract=off", "-x", "hip"],
    "mesh_file.cpp": ["-ffp-contract=off", "-x", "hip"],
    "tetsim_api.hip": ["-ffp-contract=off"],
    "tetsim_create.hip": ["-ffp-contract=off"],
    "tetsim_halo.hip": ["-ffp-contract=off"],
    "tetsim_host.cpp": ["-ffp-contract=off", "-x", "hip"],
    "pj_precise.hip": ["-ffp-contract=off"],
    "pj_fast.hip": ["-ffp-contract=fast"],
    "pj_blocked.hip": ["-ffp-contract=fast"],
    "nh_precise.hip": ["-ffp-contract=off"],
    "nh_fast.hip": ["-ffp-contract=fast"],
    "util_kernels.hip": ["-ffp-contract=off"],
    "skin_kernels.hip": ["-ffp-contract=off"],
}
HEADERS = ["body.h", "dev_common.h", "dev_store.h", "host_prep.h", "mesh_file.h", "pj_kernels.inc", "pj_math.inc", "nh_kernels.inc", os.path.join("..", "..", "include", "tetsim.h")]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(unit, flags, force):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJ, os.path.splitext(unit)[0] + ".o")
    deps = [src] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps):
        return obj, False
    cmd = [HIPCC] + COMMON + flags + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (unit, " ".join(cmd), r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(12, len(UNITS))) as ex:
        results = list(ex.map(lambda kv: _compile(kv[0], kv[1], force), UNITS.items()))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
