#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel from the CSVs tools/pmc_run.sh leaves behind.
    python tools/pmc_summary.py gpurun_out/pmc_r1 > profiles/archive/r01_pmc.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r["Kernel_Name"].replace("tetsim::(anonymous namespace)::", "").split("(")[0]
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
kernels = sorted({k for k, _ in acc})
for k in kernels:
    print(k)
    for (kk, c), (s, n) in sorted(acc.items()):
        if kk == k:
            print("    %-26s %18.1f   (mean of %d dispatches)" % (c, s / n, n))
