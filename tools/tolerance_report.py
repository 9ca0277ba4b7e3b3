#!/usr/bin/env python3
"""Table of a tolerance-calibration run:  TETSIM_RECORD_ERRORS=f.jsonl python -m pytest tests -m gpu ; python tools/tolerance_report.py f.jsonl

Per label: the largest observed error, the allowed value, their ratio, and the tolerance the <= 3 x rule suggests (observed x 3
rounded up to 1 / 2 / 5 x 10^k).  Labels whose allowed value exceeds 3 x the observed one are marked LOOSE."""
import json
import math
import sys
from collections import OrderedDict

rows = OrderedDict()
for line in open(sys.argv[1]):
    r = json.loads(line)
    k = r["label"]
    if k not in rows or r["observed"] > rows[k]["observed"]:
        rows[k] = r


def nice(x):
    if x <= 0:
        return 0.0
    e = math.floor(math.log10(x))
    for m in (1, 2, 5, 10):
        if m * 10 ** e >= x * (1 - 1e-12):
            return m * 10 ** e


print("%-78s %10s %10s %7s %10s" % ("check", "observed", "allowed", "ratio", "3x rule"))
for k, r in rows.items():
    o, a = r["observed"], r["allowed"]
    ratio = a / o if o > 0 else float("inf")
    flag = "" if o == 0 or ratio <= 3.0 + 1e-9 else "  LOOSE" if o <= a else "  FAIL"
    print("%-78s %10.3g %10.3g %7.1f %10.3g%s" % (k[:78], o, a, ratio, nice(3 * o), flag))
