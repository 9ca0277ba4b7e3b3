#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) as a per-kernel table.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_xxx_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3,"
                  " max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size),"
                  " max(workgroup_x), max(grid_x) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1.0
print("%-58s %7s %12s %9s %9s %9s %6s %5s %5s %6s %8s %5s %9s" %
      ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch", "wg", "grid_x"))
for r in rows:
    print("%-58s %7d %12.1f %9.2f %9.2f %9.2f %5.1f%% %5d %5d %6d %8d %5d %9d" %
          (r[0].replace("tetsim::(anonymous namespace)::", "")[:58], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot,
           r[6] + (r[7] or 0), r[8], r[9], r[10], r[11], r[12]))
try:
    pmc = db.execute("select k.name, p.name, avg(e.value), count(*) from pmc_events e join kernels k on k.dispatch_id = e.dispatch_id "
                     "join pmc_info p on p.id = e.pmc_id group by k.name, p.name").fetchall()
    if pmc:
        print("\ncounters (average per dispatch)")
        for r in pmc:
            print("%-58s %-24s %16.1f  n=%d" % (r[0].replace("tetsim::(anonymous namespace)::", "")[:58], r[1], r[2], r[3]))
except sqlite3.Error as e:
    print("(no counter tables: %s)" % e)
