#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the rocprofv3 --pmc passes of tools/pmc_run.sh:  python tools/pmc_traffic.py gpurun_out/<tag> [out.json]

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE counts 128-byte requests as 64 B on gfx950
(MI355X_MICROARCH.md), which the copy kernel of the same run confirms (WRITE_SIZE == bytes copied, FETCH_SIZE == half of it).
The file is keyed by the kernel_sha of the library that ran (bench.py attaches the figure only to that very build)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
acc = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = r["Kernel_Name"].replace("tetsim::(anonymous namespace)::", "").split("(")[0]
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tetsim_amd import library_info  # noqa: E402
li = library_info()
res = {"_how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh); gfx950 correction per MI355X_MICROARCH.md: "
               "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, calibrated in the same run on copy16_kernel; tools/pmc_traffic.py",
       "kernel_sha": li["kernel_sha"], "source_sha": li["source_sha"]}
for k in sorted({k for k, _ in acc}):
    if (k, "FETCH_SIZE") in acc and (k, "WRITE_SIZE") in acc and any(s in k for s in ("pjb_", "pj_", "nh_", "copy16")):
        fs, ws = acc[(k, "FETCH_SIZE")], acc[(k, "WRITE_SIZE")]
        f, w = fs[0] / fs[1], ws[0] / ws[1]
        res[k] = {"fetch_size_kb_raw": round(f, 1), "write_size_kb": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024), "dispatches": fs[1]}
with open(out, "w") as fh:
    json.dump(res, fh, indent=1)
print(json.dumps(res, indent=1))
