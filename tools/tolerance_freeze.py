#!/usr/bin/env python3
"""One-off (round 5), kept so the frozen file can be re-derived from history: tests/golden/tolerance_ceilings.json.

    python tools/tolerance_freeze.py            # prints what it would write
    python tools/tolerance_freeze.py --write

Ceiling of a label = what round 3's table (commit 57b715d) allowed it, for the 264 labels that existed then; what round 4's
final table (commit e2f084b) allowed it, for the 13 labels round 4 introduced.  Then tests/golden/tolerances.json is brought
back under the ceilings: every label round 4 loosened whose round-4 observed error still fits round 3's allowance gets round
3's allowance back (61 labels); the 4 whose observed error left it keep round 4's value WITH the reason; the two 600-substep
velocity rows of the FAST paths become report rows ("contract": false).  From here on tools/tolerance_report.py maintains
both files and never raises a ceiling."""
import json
import subprocess
import sys
from collections import OrderedDict

R3, R4 = "57b715d", "e2f084b"

REASONS = {
    "polar fast constant-rest vs carried dragon @1":
        "commit df982f7 (round 4): the carried-shape Dragon moved to the four-lane kernels of pj_quad.hip while a constant-rest-shape body "
        "keeps the one-lane 256-tet tiles, so the two sides no longer share one kernel's arithmetic (round 3: bit-equal, observed 0); "
        "4.8e-7 m is four ulps of a 1.3 m coordinate after one substep",
    "polar fast constant-rest vs carried dragon @200":
        "commit df982f7 (round 4): as '@1' -- two different kernels' re-associations instead of one, amplified over 200 substeps of contact",
    "polar fast vs reference GLSL lat4_drag @60 (vel)":
        "commit df982f7 (round 4): the 4^3 lattice is solved by the four-lane kernels (sums re-associated across a quad); a velocity is "
        "a position difference / dt with dt = 1/1200 s, so 7.2e-4 m/s is 6e-7 m of position -- the position row of the same dump did not move",
    "neo-hookean fast clustered dragon volError vs oracle frame 7":
        "commit 5172d92 (round 4): the FAST Neo-Hookean unit is built with -ffp-contract=on instead of fast (twin kernels must agree bit for "
        "bit); volError is a sum of 3,840 f32 (det F - 1) terms of order 1e-3, 3.4e-7 is rounding of that sum",
}
REPORT_WHY = ("600 substeps of contact dynamics are a chaotic horizon for a velocity (position difference / dt, dt = 1/1200 s): the row "
              "reports how far two valid f32 trajectories have drifted; parity of the arithmetic is pinned by the 1 / 20 / 200-substep rows "
              "and the position row of the same dump (VERDICT round 4, weak #1)")
REPORTS = ["polar fast vs reference GLSL dragon @600 (vel)", "polar fast gather vs reference GLSL dragon @600 (vel)"]


def table_at(rev):
    return json.loads(subprocess.check_output(["git", "show", rev + ":tests/golden/tolerances.json"]), object_pairs_hook=OrderedDict)["checks"]


def main():
    r3, r4 = table_at(R3), table_at(R4)
    ceilings = OrderedDict()
    for k, c in r4.items():
        src = r3 if k in r3 else r4
        ceilings[k] = OrderedDict(ceiling=src[k]["allowed"], since="round 3 (%s)" % R3 if k in r3 else "round 4 (%s)" % R4)
    table, restored, reasoned = OrderedDict(), 0, 0
    for k, c in r4.items():
        row = OrderedDict(observed=c["observed"], allowed=c["allowed"], stated=c["stated"])
        cap = ceilings[k]["ceiling"]
        if k in REPORTS:
            row["allowed"], row["contract"], row["why"] = c["stated"], False, REPORT_WHY
        elif c["allowed"] > cap:
            if c["observed"] <= cap:
                row["allowed"] = cap
                restored += 1
            else:
                row["reason"] = REASONS[k]
                reasoned += 1
        table[k] = row
    print("%d ceilings; %d loosened labels restored to round 3's allowance, %d kept with a reason, %d report rows"
          % (len(ceilings), restored, reasoned, len(REPORTS)))
    if "--write" in sys.argv:
        with open("tests/golden/tolerance_ceilings.json", "w") as f:
            json.dump(OrderedDict(_how=__doc__.split("\n\n")[2].replace("\n", " "), ceilings=ceilings), f, indent=1)
        how = json.load(open("tests/golden/tolerances.json"))["_how"]
        with open("tests/golden/tolerances.json", "w") as f:
            json.dump(OrderedDict(_how=how, checks=table), f, indent=1)


if __name__ == "__main__":
    main()
