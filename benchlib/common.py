"""Constants of the headline workload (BASELINE.json metric) shared by bench.py's parts."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
GOLD = os.path.join(ROOT, "tests", "golden")

SUBSTEPS = 20

CELLS = 55

PP = dict(gravity=-9.81, timeScale=1.0, timeStep=1.0 / 60.0, numSubsteps=SUBSTEPS, friction=1000.0,
          density=1000.0, devCompliance=1.0 / 100000.0, volCompliance=0.0,
          worldBounds=[-2.5, -1.0, -2.5, 2.5, 10.0, 2.5])

DT = (PP["timeScale"] * PP["timeStep"]) / PP["numSubsteps"]  # main.js:79

# SURVEY.md §8(d), reference formulation (world-space lastRest carried forward), per tet per substep:
# idx 16 R + lastRest 48 R + 48 W + quat 16 R + 16 W + restVol 4 R.
TET_KERNEL_BYTES = 148.0

# TETSIM_FLAG_LEAN_STATE (include/tetsim.h): positions 16 R + three carried corners 36 R + 36 W + restVol 4 R (no quaternion, no fourth corner)
TET_KERNEL_BYTES_LEAN = 92.0

VERTEX_BYTES = 144.0           # per particle per substep (integrate/accumulate/finalize rows)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
