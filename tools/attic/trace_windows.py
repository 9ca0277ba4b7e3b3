"""The dominant kernel per WINDOW of `bench.py --steps K --warmup W` (N = 1, polar, FAST), from a rocprofv3 --kernel-trace csv of that
command.  Windows in the order bench.py issues them (benchlib/headline.py):
  headline body:            W warm-up frames + K TIMED frames                      (graph replays, FAST exit)
  replay body:              W + K frames                                           (per-launch events, FAST exit)
  3 reference-threshold bodies: each W + K frames                                  (graph replays, |omega| < 1e-9)
  reference-threshold body: W + K frames                                           (per-launch events)
  headline body again:      180 substeps on the floor                              (per-launch events)
Per window: the tet kernel's mean duration, the particle kernel's, and the mean start-to-start interval of consecutive tet
kernels inside a frame (= what a substep takes there; interval - kernels = the two launch boundaries, or the host's gaps in the
per-launch-event windows).
python tools/attic/trace_windows.py <..._kernel_trace.csv> [K=20] [W=5] [substeps=20]"""
import csv, sys
path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
S = int(sys.argv[4]) if len(sys.argv) > 4 else 20
allrows = list(csv.DictReader(open(path)))
tet = sorted((r for r in allrows if "pjb_tet_kernel(" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
ver = sorted((r for r in allrows if "pjb_vertex_kernel(" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
ts = [int(r["Start_Timestamp"]) for r in tet]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tet]
dv = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ver]
names, edges = [], [0]
def add(n, count):
    names.append(n); edges.append(edges[-1] + count)
add("headline: warm-up frames (graph, FAST exit)", W * S)
add("headline: TIMED frames (graph, FAST exit)", K * S)
add("replay: warm-up (events, FAST exit)", W * S)
add("replay: the timed frames (events, FAST exit)", K * S)
for i in range(3):
    add("reference threshold %d: warm-up (graph)" % (i + 1), W * S)
    add("reference threshold %d: TIMED frames (graph)" % (i + 1), K * S)
add("reference threshold: warm-up (events)", W * S)
add("reference threshold: the timed frames (events)", K * S)
if edges[-1] > len(d):   # an older line (no reference-threshold leg): the first four windows and the rest
    names, edges = names[:4], edges[:5]
names.append("after the timed region, on the floor (events)"); edges.append(len(d))
print("pjb_tet_kernel, %d launches (pjb_vertex_kernel: %d) in %s" % (len(d), len(dv), path.split("/")[-1]))
print("  %-52s %-16s %9s %9s %12s" % ("window", "launches", "tet us", "vertex us", "substep us"))
for n, a, b in zip(names, edges, edges[1:]):
    if b <= a:
        continue
    # start-to-start of consecutive tet kernels, frame boundaries (every S-th) left out: between frames the host launches the next graph
    iv = [(ts[i + 1] - ts[i]) / 1e3 for i in range(a, b - 1) if (i - a + 1) % S != 0]
    mv = sum(dv[a:b]) / (b - a) if len(dv) >= b else float("nan")
    print("  %-52s %6d..%-6d %9.2f %9.2f %12.2f" % (n, a, b, sum(d[a:b]) / (b - a), mv, sum(iv) / max(len(iv), 1)))
print("  %-52s %-16s %9.2f" % ("all", "", sum(d) / len(d)))
