#!/usr/bin/env python3
"""Per-kernel launch statistics from a rocprofv3 --kernel-trace csv (the bench command of tools/ceiling_batch.sh): for every polar kernel of
the product -- the one-launch call kernels (pjb_call_kernel<0> reference record, <2> lean state: one launch = a frame of 20 substeps) and
the kernel pair tetsim_profile still runs (pjb_tet_kernel, pjb_vertex_kernel) -- launches, mean / median / min / p90 duration, and the
call kernels also per substep.  The averages here are what bench.py's roofline.rocprof quotes from the --stats summary.
    python tools/kernel_launch_stats.py <..._kernel_trace.csv> [substeps per call = 20]"""
import csv
import statistics
import sys
from collections import defaultdict

path = sys.argv[1]
S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dur = defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if not any(k in n for k in ("pjb_", "nh_sweep1", "nh_cluster4")):
        continue
    key = n.replace("void ", "").replace("tetsim::(anonymous namespace)::", "").split("(")[0].replace("(int)", "")
    dur[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("%-44s %8s %10s %10s %10s %10s %12s" % ("kernel", "launches", "mean us", "median us", "min us", "p90 us", "per substep"))
for k in sorted(dur):
    d = [x[1] for x in sorted(dur[k])]
    per = "%.2f us" % (statistics.mean(d) / S) if "call_kernel" in k else ""
    print("%-44s %8d %10.2f %10.2f %10.2f %10.2f %12s" % (k[:44], len(d), statistics.mean(d), statistics.median(d), min(d), sorted(d)[int(0.9 * (len(d) - 1))], per))
