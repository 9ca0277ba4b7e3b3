"""Builds libtetsim_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m tetsim_amd.build [--force] [--ablation]

Per-file flags matter: the PRECISE translation units and the host preprocessing reproduce the reference's
rounding and must not fuse multiply-add (-ffp-contract=off); the FAST units are built with contraction on.

--ablation additionally builds libtetsim_hip_ablation.so: the same sources with -DTETSIM_ABLATION, whose polar tet kernel
takes the timing-ablation knobs (TETSIM_DEBUG_ITERS / _SKIP_REST_STORE / _NO_PEEL) and the per-tile trace.  Development
only (tools/attic/ab_iters.py, tools/attic/trace_tet.py); the product library has none of that code.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtetsim_hip.so")
LIB_ABLATION = os.path.join(HERE, "libtetsim_hip_ablation.so")
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
    "partitioner.cpp": ["-ffp-contract=off", "-x", "hip"],
    "tetsim_api.hip": ["-ffp-contract=off"],
    "tetsim_state.hip": ["-ffp-contract=off"],
    "tetsim_visual.hip": ["-ffp-contract=off"],
    "tetsim_measure.hip": ["-ffp-contract=off"],
    "tetsim_create.hip": ["-ffp-contract=off"],
    "tetsim_halo.hip": ["-ffp-contract=off"],
    "tetsim_comm.hip": ["-ffp-contract=off"],
    "tetsim_p2p.hip": ["-ffp-contract=off"],
    "tetsim_host.cpp": ["-ffp-contract=off", "-x", "hip"],
    "pj_precise.hip": ["-ffp-contract=off"],
    "pj_fast.hip": ["-ffp-contract=fast"],
    # on, not fast (see nh_fast.hip below): the per-substep, fused and persistent kernels of this unit share their arithmetic and are held
    # to bit-equality; the multiply-adds that matter are spelled out.  In-run A/B on the 1 M-tet lattice: tet kernel 26.8-27.1 us either way.
    "pj_blocked.hip": ["-ffp-contract=on"],
    # four lanes per tet, small bodies: the frame kernel and its two-launch substep must agree bit for bit, so no contraction is left to
    # the compiler -- every fused multiply-add is spelled out in the source
    "pj_quad.hip": ["-ffp-contract=off"],
    "nh_precise.hip": ["-ffp-contract=off"],
    # kernarg preload: the four-lane cluster kernel's leading scalar arguments (ids, particles, mask, count) arrive in SGPRs with the
    # wave instead of through a scalar load at the head of every colour's chain: 66.8 -> 65.4 us per substep (profiles/archive/r03_neohookean.txt)
    # contract=on, not fast: with fast the BACKEND fuses any multiply with any add it finds, whatever `#pragma clang fp contract` says, and
    # picks the pairs by the surrounding code -- kernels that share their arithmetic (a body's fused launch and its stepwise twin) then
    # round differently in one tet out of a few hundred.  on = only a * b + c written as one expression (or fmaf) fuses.
    "nh_fast.hip": ["-ffp-contract=on", "-mllvm", "-amdgpu-kernarg-preload-count=9"],
    "util_kernels.hip": ["-ffp-contract=off"],
    "skin_kernels.hip": ["-ffp-contract=off"],
    "build_info.cpp": ["-x", "hip"],
}
HEADERS = ["body.h", "dev_common.h", "dev_store.h", "host_prep.h", "mesh_file.h", "pj_kernels.inc", "pj_math.inc", "pj_lab.h", "pj_blocked_lab.inc", "nh_kernels.inc", os.path.join("..", "..", "include", "tetsim.h")]
# what determines the polar tet kernel (pjb_tet_kernel), its tiling and therefore its HBM traffic: profiles/pmc_traffic.json
# is keyed by the hash of these (+ their flags), so a stale counter figure is never attached to a different kernel
KERNEL_FILES = ["pj_blocked.hip", "pj_math.inc", "pj_lab.h", "dev_common.h", "dev_store.h", "host_prep.cpp", "host_prep.h"]


def _sha(files, extra):
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(repr(extra).encode())
    return h.hexdigest()[:16]


def source_shas():
    """(source_sha, kernel_sha) of the tree as it is now -- what a library built now would report."""
    every = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".inc")) and f != "build_info.cpp") + \
        [os.path.join("..", "..", "include", "tetsim.h")]
    flags = (COMMON, sorted(UNITS.items()))
    return _sha(every, flags), _sha(KERNEL_FILES, (COMMON, UNITS["pj_blocked.hip"], UNITS["host_prep.cpp"]))


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(unit, flags, force, objdir, extra):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(objdir, os.path.splitext(unit)[0] + ".o")
    deps = [src] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    stamp = obj + ".flags"
    want = " ".join(flags + extra)
    same_flags = os.path.exists(stamp) and open(stamp).read() == want
    if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps):
        return obj, False
    cmd = [HIPCC] + COMMON + flags + extra + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (unit, " ".join(cmd), r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(want)
    return obj, True


def _build_one(lib, objdir, defines, force, verbose, unit_flags=None):
    os.makedirs(objdir, exist_ok=True)
    src_sha, ker_sha = source_shas()
    info = ['-DTETSIM_SOURCE_SHA="%s"' % src_sha, '-DTETSIM_KERNEL_SHA="%s"' % ker_sha]
    unit_flags = unit_flags or {}

    def one(kv):
        unit, flags = kv
        return _compile(unit, flags + unit_flags.get(unit, []), force, objdir, defines + (info if unit == "build_info.cpp" else []))

    with ThreadPoolExecutor(max_workers=min(12, len(UNITS))) as ex:
        results = list(ex.map(one, UNITS.items()))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(lib):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose:
            print("linked", lib)
    return lib


def build(force=False, verbose=False, ablation=False):
    lib = _build_one(LIB, os.path.join(CSRC, "obj"), [], force, verbose)
    if ablation:
        _build_one(LIB_ABLATION, os.path.join(CSRC, "obj_ablation"), ["-DTETSIM_ABLATION"], force, verbose)
    return lib


def build_variant(name, defines, force=False, verbose=False, unit_flags=None):
    """Development: libtetsim_hip_<name>.so with extra -D flags (kernel experiments, A/B through TETSIM_HIP_LIB; tools/ab_lib.py);
    unit_flags = {"pj_blocked.hip": ["-fslp-vectorize"]} appends compiler flags to single units (--unit-flag unit=flag)."""
    return _build_one(os.path.join(HERE, "libtetsim_hip_%s.so" % name), os.path.join(CSRC, "obj_" + name), list(defines), force, verbose, unit_flags)


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m tetsim_amd.build --variant t128 -DTETSIM_TILE=128
        i = sys.argv.index("--variant")
        uf = {}
        for j, a in enumerate(sys.argv):
            if a == "--unit-flag":   # --unit-flag pj_blocked.hip=-fslp-vectorize
                unit, flag = sys.argv[j + 1].split("=", 1)
                uf.setdefault(unit, []).append(flag)
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")], force="--force" in sys.argv, verbose=True, unit_flags=uf))
    else:
        print(build(force="--force" in sys.argv, verbose=True, ablation="--ablation" in sys.argv))
