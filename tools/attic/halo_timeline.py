"""Timeline of the two-queue halo choreography from a rocprofv3 --kernel-trace CSV (tools/loopback_rank.py under the profiler):
for a few substeps in the middle of the run, every kernel with its queue, start and end relative to the first one (us)."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
def short(n):
    for k, v in (("pjb_tet", "TET"), ("pjb_vertex", "PART"), ("pjb_wait", "wait"), ("pjb_signal", "signal"), ("delay", "delay"), ("nccl", "RCCL"), ("gather16", "pack"), ("copy16", "copy"), ("repredict", "repredict")):
        if k in n: return v
    return n[:24]
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 else 3
# the last call of the run: find the last 20 * n kernels ... simpler: take a window that starts at a TET kernel 60% into the trace
i0 = int(len(rows) * 0.6)
while "pjb_tet" not in rows[i0][2]: i0 += 1
t0 = rows[i0][0]
tets = 0
for s, e, n, q, st in rows[i0:]:
    if "pjb_tet" in n: tets += 1
    if tets > 2 * n_sub: break
    print("%8.1f .. %8.1f  (%5.1f us)  queue %-3s stream %-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, short(n)))
