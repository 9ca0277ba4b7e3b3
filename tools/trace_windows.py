"""The dominant kernel's mean duration per WINDOW of `bench.py --steps K --warmup W`, from a rocprofv3 --kernel-trace csv of that command:
warm-up frames | timed frames (graph replays) | the replay's warm-up | the replay of the timed frames (per-launch events) | the window after.
python tools/trace_windows.py <..._kernel_trace.csv> [K=20] [W=5] [substeps=20]"""
import csv, sys
path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
W = int(sys.argv[3]) if len(sys.argv) > 3 else 5
S = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rows = [r for r in csv.DictReader(open(path)) if "pjb_tet_kernel(" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
edges = [0, W * S, (W + K) * S, (2 * W + K) * S, (2 * W + 2 * K) * S, len(d)]
names = ["warm-up frames (graph replays)", "TIMED frames (graph replays)", "replay: warm-up (per-launch events)", "replay: the timed frames (per-launch events)", "after the timed region (per-launch events)"]
print("pjb_tet_kernel, %d launches in %s" % (len(d), path.split("/")[-1]))
for n, a, b in zip(names, edges, edges[1:]):
    if b > a:
        print("  %-48s launches %4d..%4d  mean %.2f us" % (n, a, b, sum(d[a:b]) / (b - a)))
print("  %-48s %s mean %.2f us" % ("all", " " * 20, sum(d) / len(d)))
