"""The drop-in boundary from plain C (examples/lattice_demo.c): include/tetsim.h is a C header, libtetsim_hip.so links into a C99
program, and that program fails loudly where there is no GPU (no CPU path) and reproduces the Python host's result where there is one."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    from tetsim_amd import build
    build.build()
    exe = str(tmp_path_factory.mktemp("cdemo") / "lattice_demo")
    lib_dir = os.path.join(ROOT, "tetsim_amd")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "lattice_demo.c"),
           "-L" + lib_dir, "-ltetsim_hip", "-Wl,-rpath," + lib_dir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr          # the header is valid, warning-free C99 and every symbol the program uses is exported
    return exe


def test_the_c_program_fails_loudly_without_a_gpu(demo):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU test below runs the program to the end")
    r = subprocess.run([demo, "4", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and r.stdout == ""


def test_the_c_lattice_is_the_python_lattice():
    """The example's own generator (double arithmetic, rounded once) and tetsim_amd.make_lattice agree -- checked through the C source's
    constants: six axis orders in itertools.permutations order, vertex index i + m (j + m k), corners 2 and 3 swapped where left-handed."""
    src = open(os.path.join(ROOT, "examples", "lattice_demo.c")).read()
    perms = re.search(r"perm\[6\]\[3\] = \{(.*?)\};", src).group(1)
    import itertools
    assert [tuple(int(x) for x in g) for g in re.findall(r"\{(\d), (\d), (\d)\}", perms)] == list(itertools.permutations(range(3)))


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["polar", "neohookean"])
def test_the_c_program_reproduces_the_python_host(demo, solver):
    from tetsim_amd import SoftBodyHIP, make_lattice
    cells, frames = 8, 5
    r = subprocess.run([demo, str(cells), str(frames), solver], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.search(r"(\d+) tets (\d+) particles .* sum ([-0-9.e+]+) ymin ([-0-9.e+]+)", r.stdout)
    v, t = make_lattice(cells, y0=0.05)
    assert (int(m.group(1)), int(m.group(2))) == (len(t), len(v))
    b = SoftBodyHIP(v, t, None, {}, solver=solver, precision="fast", order="clustered")      # tetsim_default_options / _params, as the C program
    dt = (1.0 * (1.0 / 60.0)) / 20
    for _ in range(frames):
        b.simulateSubsteps(20, dt, None)
    pos = b.pos.astype(np.float64)
    s = 0.0
    for x, y, z in pos:
        s += x + y + z
    assert float("%.9g" % s) == float(m.group(3)) and float("%.6g" % pos[:, 1].min()) == float(m.group(4))
    assert float(m.group(4)) >= 0.0 and float(m.group(4)) < 0.05          # it has reached the floor
