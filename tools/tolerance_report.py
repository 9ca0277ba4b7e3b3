#!/usr/bin/env python3
"""Table of a tolerance-calibration run, and the committed calibration file.

    TETSIM_RECORD_ERRORS=$PWD/errors.jsonl python -m pytest tests -m gpu     (on the GPU box: checks record instead of failing)
    python tools/tolerance_report.py errors.jsonl [--write tests/golden/tolerances.json]

Per label: the largest observed error, the bound the test states, and 3 x observed (rounded DOWN to three significant digits) --
the value tests/conftest.py:within() enforces once written.  Labels whose observed error exceeds the stated bound are marked FAIL."""
import json
import math
import sys
from collections import OrderedDict


def three_times(x):
    """3 x the observed error, rounded DOWN to three significant digits (never above 3 x)."""
    if x <= 0:
        return 0.0
    y = 3.0 * x
    e = math.floor(math.log10(y)) - 2
    return float("%.6g" % (math.floor(y / 10 ** e + 1e-9) * 10 ** e))


def ulp_bound(label):
    """The bound for a check whose calibrated error is exactly 0 (results bit-identical to the oracle / the reference's own
    output): one to two ulps of the quantity -- unit quaternions 1.2e-7, positions of a metre-sized body 2.5e-7, velocities of
    a few m/s 1e-6 -- so that anything systematic, however small, shows."""
    return 1.2e-7 if "(quat)" in label else 1e-6 if "(vel)" in label else 2.5e-7


def main():
    rows = OrderedDict()
    for line in open(sys.argv[1]):
        r = json.loads(line)
        k = r["label"]
        if k not in rows or r["observed"] > rows[k]["observed"]:
            rows[k] = r
    print("%-84s %10s %10s %10s" % ("check", "observed", "stated", "3x observed"))
    for k, r in rows.items():
        o, a = r["observed"], r["allowed"]
        print("%-84s %10.3g %10.3g %10.3g%s" % (k[:84], o, a, three_times(o), "  FAIL (above the stated bound)" if o > a else ""))
    if "--write" in sys.argv:
        path = sys.argv[sys.argv.index("--write") + 1]
        out = {"_how": "tools/tolerance_report.py from a TETSIM_RECORD_ERRORS calibration run of `pytest -m gpu` on MI355X; "
                       "allowed = 3 x observed rounded down to 3 digits, capped by the bound the test states; observed 0 (bit-identical) -> an ulp-level bound "
                       "(quaternions 1.2e-7, positions 2.5e-7, velocities 1e-6)",
               "checks": OrderedDict((k, {"observed": r["observed"], "allowed": min(three_times(r["observed"]), r["allowed"]) if r["observed"] > 0 else min(ulp_bound(k), r["allowed"]),
                                          "stated": r["allowed"]}) for k, r in rows.items())}
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", path, "(%d checks)" % len(rows))


if __name__ == "__main__":
    main()
