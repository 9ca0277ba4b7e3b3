#!/usr/bin/env python3
"""Table of a tolerance-calibration run, and the committed calibration file -- which only ever ratchets DOWN.

    TETSIM_RECORD_ERRORS=$PWD/errors.jsonl python -m pytest tests -m gpu     (on the GPU box: checks record instead of failing)
    python tools/tolerance_report.py errors.jsonl [--write] [--reason 'LABEL=commit abc1234: why' ...] [--reasons FILE.json]

Two files under tests/golden/:

* tolerance_ceilings.json -- the FROZEN contract: per label the largest error the suite has ever been allowed to accept
  (round 3's table for the labels that existed then, the table of the round that introduced it for a later one).  This tool
  adds ceilings for labels it has never seen; it never raises one.  tests/test_capi_cpu.py holds its digest.
* tolerances.json -- per label the error observed in the last calibration run and what tests/conftest.py:within() enforces:
  allowed = min(3 x observed rounded down, the bound the test states, the label's ceiling).  A label may be allowed more than
  its ceiling ONLY with a "reason" string that names the commit and the cause; without one a calibration run that observes
  more than a ceiling is REFUSED (exit 2, nothing written): the kernel drifted, fix it or justify it.
  Rows with "contract": false are reports (chaotic horizons: the number says how far two valid trajectories have drifted,
  not whether the arithmetic is right); only the bound their test states applies to them.

Labels whose observed error exceeds the stated bound are marked FAIL."""
import json
import math
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "tolerances.json")
CEILINGS = os.path.join(ROOT, "tests", "golden", "tolerance_ceilings.json")


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


def calibrated(label, observed, stated):
    return min(three_times(observed), stated) if observed > 0 else min(ulp_bound(label), stated)


def load_json(path, default):
    try:
        with open(path) as f:
            return json.load(f, object_pairs_hook=OrderedDict)
    except FileNotFoundError:
        return default


def merge(rows, table, ceilings, reasons, round_tag):
    """The new table and ceilings from a calibration run.  Returns (table, ceilings, refused): `refused` lists the labels that
    observed more than their ceiling and carry no reason."""
    out, refused = OrderedDict(), []
    for k, r in rows.items():
        o, stated = r["observed"], r["allowed"]
        old = table.get(k, {})
        want = calibrated(k, o, stated)
        row = OrderedDict(observed=o, allowed=want, stated=stated)
        if old.get("contract") is False or r.get("contract") is False:
            row["allowed"], row["contract"] = stated, False
            row["why"] = old.get("why", "chaotic horizon: a report of drift between two valid trajectories, not a parity bound")
            out[k] = row
            continue
        if k not in ceilings:
            ceilings[k] = OrderedDict(ceiling=want, since=round_tag)
        cap = ceilings[k]["ceiling"]
        reason = reasons.get(k) or old.get("reason")
        if o > cap and not reason:
            refused.append((k, o, cap))
        if want > cap:
            if reason and o > cap:
                row["reason"] = reason          # above the ceiling, and says why
            else:
                row["allowed"] = cap            # 3 x observed would loosen the contract: the ceiling stays
        out[k] = row
    for k, old in table.items():                # a partial run keeps the labels it did not exercise
        out.setdefault(k, old)
    return out, ceilings, refused


def main():
    rows = OrderedDict()
    for line in open(sys.argv[1]):
        r = json.loads(line)
        k = r["label"]
        if k not in rows or r["observed"] > rows[k]["observed"]:
            rows[k] = r
    table = load_json(TABLE, {"checks": {}})["checks"]
    cfile = load_json(CEILINGS, {"ceilings": OrderedDict()})
    ceilings = cfile["ceilings"]
    reasons = {}
    args = sys.argv[2:]
    for i, a in enumerate(args):
        if a == "--reason":
            k, _, why = args[i + 1].partition("=")
            reasons[k] = why
        if a == "--reasons":
            reasons.update(json.load(open(args[i + 1])))
    round_tag = next((args[i + 1] for i, a in enumerate(args) if a == "--round"), "round 6")
    print("%-84s %10s %10s %10s %10s" % ("check", "observed", "stated", "3x observed", "ceiling"))
    for k, r in rows.items():
        o, a = r["observed"], r["allowed"]
        cap = ceilings.get(k, {}).get("ceiling", float("nan"))
        print("%-84s %10.3g %10.3g %10.3g %10.3g%s%s" % (k[:84], o, a, three_times(o), cap,
                                                       "  FAIL (above the stated bound)" if o > a else "",
                                                       "  ABOVE ITS CEILING" if o > cap and table.get(k, {}).get("contract") is not False else ""))
    new_table, ceilings, refused = merge(rows, table, ceilings, reasons, round_tag)
    if refused:
        print("\nREFUSED: %d label(s) observed more than their frozen ceiling and carry no reason:" % len(refused))
        for k, o, cap in refused:
            print("  %-84s observed %.3g > ceiling %.3g" % (k[:84], o, cap))
        print("fix the drift, or pass --reason 'LABEL=commit <sha>: <cause>' for each")
        sys.exit(2)
    if "--write" in args:
        how = ("tools/tolerance_report.py from a TETSIM_RECORD_ERRORS calibration run of `pytest -m gpu` on MI355X; allowed = 3 x observed "
               "rounded down to 3 digits, capped by the bound the test states AND by the label's frozen ceiling "
               "(tolerance_ceilings.json) unless a \"reason\" names the commit and the cause; observed 0 (bit-identical) -> an "
               "ulp-level bound (quaternions 1.2e-7, positions 2.5e-7, velocities 1e-6); \"contract\": false = a report row")
        with open(TABLE, "w") as f:
            json.dump({"_how": how, "checks": new_table}, f, indent=1)
        cfile["ceilings"] = ceilings
        with open(CEILINGS, "w") as f:
            json.dump(cfile, f, indent=1)
        print("wrote", TABLE, "(%d checks) and" % len(new_table), CEILINGS, "(%d ceilings)" % len(ceilings))


if __name__ == "__main__":
    main()
