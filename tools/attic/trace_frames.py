"""Per-FRAME means of the tet kernel, the particle kernel and the substep (start-to-start of consecutive tet kernels) from a rocprofv3
--kernel-trace csv of `bench.py --steps K --warmup W`: the headline body's W + K frames (graph replays, FAST exit), then the first
reference-threshold body's (graph replays) -- how the kernel's duration moves with the phase of the fall, inside the graphs.
python tools/attic/trace_frames.py <..._kernel_trace.csv> K W [every=10]"""
import csv, sys
path, K, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
S = 20
rows = list(csv.DictReader(open(path)))
tet = sorted((r for r in rows if "pjb_tet_kernel(" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
ver = sorted((r for r in rows if "pjb_vertex_kernel(" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
ts = [int(r["Start_Timestamp"]) for r in tet]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tet]
dv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ver]
F = W + K
def series(name, first):
    print(name)
    print("  %-12s %9s %9s %11s" % ("frames", "tet us", "vertex us", "substep us"))
    for f0 in range(0, F, every):
        a, b = first + f0 * S, first + min(f0 + every, F) * S
        iv = [(ts[i + 1] - ts[i]) / 1e3 for i in range(a, b - 1) if (i - first + 1) % S != 0]
        print("  %4d..%-6d %9.2f %9.2f %11.2f" % (f0, min(f0 + every, F), sum(d[a:b]) / (b - a), sum(dv[a:b]) / (b - a), sum(iv) / len(iv)))
series("headline body, graph replays, FAST exit (frame 0 = first warm-up frame; the body reaches the floor around frame 19)", 0)
series("first reference-threshold body, graph replays (|omega| < 1e-9)", 2 * F * S)
