#!/usr/bin/env python3
"""What bounds pjb_tet_kernel on the 1 M-tet lattice -- the closing table VERDICT round 4 (item 3) asks for.

    python tools/tet_kernel_ceiling.py [--pmc gpurun_out/<tag>/pmc] [--iters gpurun_out/<tag>/ab_iters.txt] [--write]

1. ISA instruction histogram per phase of the PRODUCT kernel (hipcc -S of pj_blocked.hip with the product's flags; phases cut at
   the kernel's barriers and at the rotation loop): VALU (plain / transcendental), SALU, LDS, global loads / stores -- static
   counts, and dynamic counts per wave for nine iterations (the loop body x 8 + the peeled first iteration).
2. Vector-issue floor: SQ_INSTS_VALU per launch (counters, tools/pmc_run.sh) / 1,024 SIMDs x the issue rates this chip sustains with
   several waves per SIMD (profiles/archive/r01e_valu_issue_rates.txt: 3.15 cycles per plain, 8.8 per transcendental wave-instruction) at
   2.4 GHz; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES and SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES beside it.
3. Memory floor: the PRODUCT kernel rebuilt with 0 / 3 / 6 rotation iterations (tools/iteration_floor.sh; the development build's
   run-time knob compiles to a slower kernel and cannot give the product's floor).
--write: profiles/tet_kernel_ceiling.json (keyed by kernel_sha; bench.py attaches it as roofline.ceiling)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PLAIN_CYCLES, TRANS_CYCLES, CLOCK_GHZ, SIMDS = 3.15, 8.8, 2.4, 1024
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_", "v_exp_", "v_log_")


def classify(op):
    if op.startswith(TRANS):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_rd"
    if op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic", "buffer_atomic")):
        return "vmem_wr"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_isa():
    from tetsim_amd import build
    flags = build.COMMON + build.UNITS["pj_blocked.hip"]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "pjb.s")
        cmd = [build.HIPCC] + flags + ["--cuda-device-only", "-S", "-o", out, os.path.join(build.CSRC, "pj_blocked.hip")]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
    m = re.search(r"^(_ZN6tetsim12_GLOBAL__N_114pjb_tet_kernelENS_5PJBlkEjjj):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M)
    body = m.group(2).split("\n")
    meta = re.search(r"; NumVgprs: (\d+)", text[m.end():])
    return body, int(meta.group(1)) if meta else None


def phases(lines):
    """(phase, label or instruction) in program order.  Phases: stage (up to the first barrier), solve-head (up to the loop header),
    loop (the 'Inner Loop Header' block up to its back edge incl. the blocks `in Loop`), solve-tail (up to the second barrier), reduce."""
    out, phase, barriers, in_loop = [], "1 stage (ids, tet record, position gather -> LDS)", 0, False
    for ln in lines:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith("."):
            if barriers != 1:        # (the reduce phase has loops of its own: the per-particle entry lists)
                continue
            if "Loop Header" in s or "in Loop:" in s:
                in_loop = True
                phase = "3 rotation loop (one iteration)"
            elif s.startswith(".LBB") and in_loop and "Loop" not in s:
                in_loop = False
                phase = "4 solve tail (normalise, goals, LDS planes, result stores)"
            continue
        op = s.split()[0]
        if op == "s_barrier":
            barriers += 1
            out.append((phase, op))
            phase = "2 solve head (corners from LDS, covariance, peeled iteration 1)" if barriers == 1 else "5 reduce (per tile particle) + partial-sum store"
            continue
        out.append((phase, op))
    return out


def main():
    args = sys.argv[1:]
    lines, vgprs = kernel_isa()
    hist = {}
    for ph, op in phases(lines):
        hist.setdefault(ph, {}).setdefault(classify(op), 0)
        hist[ph][classify(op)] += 1
    cols = ["valu", "valu_trans", "salu", "smem", "lds", "vmem_rd", "vmem_wr", "waitcnt", "barrier"]
    print("pjb_tet_kernel (product build, %s VGPRs): static instruction counts per phase" % vgprs)
    print("%-72s" % "phase" + "".join("%11s" % c for c in cols))
    for ph in sorted(hist):
        print("%-72s" % ph + "".join("%11d" % hist[ph].get(c, 0) for c in cols))
    loop = hist.get("3 rotation loop (one iteration)", {})
    stat_valu = sum(h.get("valu", 0) for h in hist.values())
    stat_trans = sum(h.get("valu_trans", 0) for h in hist.values())
    dyn_valu = stat_valu + 7 * loop.get("valu", 0)
    dyn_trans = stat_trans + 7 * loop.get("valu_trans", 0)
    print("dynamic, nine iterations (loop body x 8): %d plain + %d transcendental VALU per wave = %.0f issue cycles of its SIMD at %.2f / %.1f cycles"
          % (dyn_valu, dyn_trans, dyn_valu * PLAIN_CYCLES + dyn_trans * TRANS_CYCLES, PLAIN_CYCLES, TRANS_CYCLES))
    res = {"_how": "tools/tet_kernel_ceiling.py", "isa": {"vgprs": vgprs, "per_phase": hist, "dynamic_plain_valu_per_wave": dyn_valu, "dynamic_trans_valu_per_wave": dyn_trans}}
    trans_share = dyn_trans / float(dyn_valu + dyn_trans)

    if "--pmc" in args:
        import csv
        import glob
        from collections import defaultdict
        acc = defaultdict(lambda: [0.0, 0])
        for f in sorted(glob.glob(os.path.join(args[args.index("--pmc") + 1], "**", "*counter_collection.csv"), recursive=True)):
            for r in csv.DictReader(open(f)):
                if "::pjb_tet_kernel(" in r["Kernel_Name"]:
                    a = acc[r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
        c = {k: s / n for k, (s, n) in acc.items()}
        per_simd = c["SQ_INSTS_VALU"] / SIMDS
        cyc = per_simd * ((1 - trans_share) * PLAIN_CYCLES + trans_share * TRANS_CYCLES)
        valu_floor = cyc / (CLOCK_GHZ * 1e3)
        print("\ncounters per launch (mean over %d dispatches of the bench command: falling and on-the-floor frames mixed):" % acc["SQ_INSTS_VALU"][1])
        for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"):
            if k in c:
                print("    %-22s %16.1f" % (k, c[k]))
        print("  VALU wave-instructions per wave            %8.1f   (static ISA count for nine iterations: %d)" % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"], dyn_valu + dyn_trans))
        print("  VALU wave-instructions per SIMD            %8.0f   -> %.0f issue cycles (%.1f%% transcendental) = %.1f us at %.1f GHz: the VECTOR-ISSUE FLOOR"
              % (per_simd, cyc, 100 * trans_share, valu_floor, CLOCK_GHZ))
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES is summed over the chip's 32 shader engines, SQ_ACTIVE_INST_VALU over its 1,024 SIMDs in units of four cycles:
            # per SIMD, the share of the kernel's busy time in which it issues a vector instruction = ACTIVE x 4 / (1,024 x BUSY / 32)
            print("  SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES       %8.3f   -> a SIMD issues vector instructions in %.0f%% of its shader engine's busy cycles (x 4 cycles / 32 SIMDs per engine)"
                  % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"], 100.0 * c["SQ_ACTIVE_INST_VALU"] * 4.0 / (SIMDS * c["SQ_BUSY_CYCLES"] / 32.0)))
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
            print("  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES       %8.3f   (share of a resident wave's life spent issuing vector instructions)" % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
            print("  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES          %8.3f   (share spent waiting for an instruction to issue or return: memory, LDS, a busy SIMD)" % (c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))
        res["counters"] = {k: round(v, 1) for k, v in c.items()}
        res["valu_issue_floor_us"] = round(valu_floor, 2)
    if "--iters" in args:
        it = {}
        for ln in open(args[args.index("--iters") + 1]):
            m = re.match(r"iters=(\d+)\s+tet ([0-9.]+) us", ln)
            if m:
                it[int(m.group(1))] = float(m.group(2))
        if it:
            print("\niteration ablation (the product kernel rebuilt with fewer iterations; on the floor, per-launch events): " + "  ".join("%d: %.2f us" % kv for kv in sorted(it.items())))
            res["iteration_ablation_us"] = it
            res["memory_floor_us"] = it.get(0)
            if 9 in it and 3 in it:
                print("  memory floor (0 iterations) %.2f us; iterations 1-3 hide under it (%.2f us at 3); 4-9 cost %.2f us each" % (it[0], it[3], (it[9] - it[3]) / 6.0))
    if "--write" in args and "memory_floor_us" in res and "valu_issue_floor_us" in res:
        from tetsim_amd import build
        src_sha, ker_sha = build.source_shas()
        out = {"_how": "tools/tet_kernel_ceiling.py on an MI355X box: memory floor = the PRODUCT kernel rebuilt with zero rotation iterations (tools/iteration_floor.sh), on the floor, per-launch events; "
                       "vector-issue floor = SQ_INSTS_VALU per SIMD x measured issue cycles (3.15 plain, 8.8 transcendental) at 2.4 GHz",
               "kernel_sha": ker_sha, "memory_floor_us": res["memory_floor_us"], "valu_issue_floor_us": res["valu_issue_floor_us"],
               "valu_instructions_per_wave": round(res["counters"]["SQ_INSTS_VALU"] / res["counters"]["SQ_WAVES"], 1),
               "source": "profiles/r06z_tet_kernel_ceiling.txt"}
        i = args.index("--write")
        path = args[i + 1] if i + 1 < len(args) and not args[i + 1].startswith("--") else os.path.join(ROOT, "profiles", "tet_kernel_ceiling.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        print("\nwrote", path)


if __name__ == "__main__":
    main()
