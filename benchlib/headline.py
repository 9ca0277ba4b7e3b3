"""One rank of the benchmark: the body, the timed region (W untimed + K timed frames bracketed by synchronise + barrier), the retry
ladder of an N-rank run, and the JSON line with its roofline object."""
import json
import os
import sys
import time

import numpy as np

from .common import CELLS, DT, HBM_PEAK_GBS, PP, ROOT, SUBSTEPS, TET_KERNEL_BYTES, TET_KERNEL_BYTES_LEAN, VERTEX_BYTES
from .cpu import cpu_baseline
from .body import make_body, timed_frames
from .launcher import GUARD
from .legs import beyond_mall, demote_p2p, halo_probe_report, multi_gpu_report, other_configs, p2p_check, promote_p2p, run_neohookean, transport_check


# How an N-rank headline run may be repeated when a transport fails on the node it meets: each rung rebuilds every rank's body with
# more conservative halo settings (the library reads them per body).  A rung is left only by a VOTE of all ranks, and every rank
# runs the same collectives whether its local steps worked or not.
HALO_LADDER = [({}, "flag-synchronised two-queue halo, both chains replayed from captured graphs (the default)"),
               ({"TETSIM_HALO_GRAPH": "0"}, "the same halo path enqueued eagerly (no graph replay)"),
               ({"TETSIM_HALO_SYNC": "events", "TETSIM_HALO_GRAPH": "0"}, "event-synchronised halo path, eager (round 1's)")]

P2P_RUNG = "peer-to-peer halo (the transport an N-rank run LEADS with): boundary particles stored straight into the neighbours' IPC-mapped ghost ranges, " \
           "flag-synchronised two-queue path replayed from captured graphs"


def headline_with_retries(args, cells, rank, world, local_rank, ranks):
    """The timed region of an N-rank run (timed_frames' protocol: W untimed + K timed frames bracketed by synchronise + barrier), with
    every local step caught and voted on.  The peer-to-peer halo goes first (--halo p2p, the default: DESIGN.md 7 -- RCCL's grouped
    send / recv kernel alone is 11-13 us of the halo queue's 30 us cycle); on a failure anywhere all ranks close their bodies and climb
    down: RCCL with the default settings, then HALO_LADDER's more conservative ones.  Thread-ranks of ONE process (--fake-ranks) cannot
    step a peer-to-peer halo independently (include/tetsim.h: tetsim_halo_p2p_connect) -- that rung is recorded as skipped.
    Returns (body, verts, tets, pp, nz, wall seconds of this rank, host seconds inside the K calls, [attempt records], transport used)."""
    attempts = []
    keys = sorted({k for env, _ in HALO_LADDER for k in env} | {"TETSIM_HALO_TIMEOUT_MS"})
    saved = {k: os.environ.get(k) for k in keys}
    ladder = ([(args.halo, {}, P2P_RUNG + (" over a two-layer ghost region" if args.halo == "deep" else ""))] if args.halo in ("p2p", "deep") else []) + \
             [("rccl", env, what) for env, what in HALO_LADDER]
    tried = 0
    try:
        for halo, env, what in ladder:
            if halo != "rccl" and args.fake_ranks:
                attempts.append({"halo": what, "ok": False, "skipped": "thread-ranks of one process cannot step a peer-to-peer halo independently (tetsim_group_step_n would)"})
                continue
            for k in keys:
                if k != "TETSIM_HALO_TIMEOUT_MS":
                    os.environ.pop(k, None) if saved[k] is None else os.environ.__setitem__(k, saved[k])
            os.environ.update(env)
            if saved["TETSIM_HALO_TIMEOUT_MS"] is None:
                os.environ["TETSIM_HALO_TIMEOUT_MS"] = "10000"   # a rank that waits in vain says so after 10 s, not 30
            state = {"err": None}

            def local(fn):
                if state["err"] is None:
                    try:
                        return fn()
                    except Exception as e:  # noqa: BLE001
                        state["err"] = "rank %d: %r" % (rank, e)
                return None

            body, verts, tets, pp, nz, err = make_body(args, cells, args.scaling, rank, world, local_rank, ranks, vote=True, halo=halo)
            el = host = 0.0
            if body is None:
                state["err"] = err
            else:
                for _ in range(args.warmup):
                    local(lambda: body.simulateSubsteps(SUBSTEPS, DT, pp))
                local(body.sync)
                ranks.barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    h0 = time.perf_counter()
                    local(lambda: body.simulateSubsteps(SUBSTEPS, DT, pp))
                    host += time.perf_counter() - h0
                local(body.sync)
                ranks.barrier()
                el = time.perf_counter() - t0
                fin = local(lambda: bool(np.isfinite(body.pos).all()))
                if fin is False:
                    state["err"] = "rank %d: non-finite positions after the timed region" % rank
                if tried == 0 and os.environ.get("TETSIM_BENCH_TEST_FAIL_FIRST_RUNG") == str(rank) and not state["err"]:
                    state["err"] = "rank %d: injected failure (test of the retry ladder)" % rank
            tried += 1
            ok = ranks.min_float(0.0 if state["err"] else 1.0) >= 1.0
            attempts.append({"halo": what, "ok": ok} if ok or not state["err"] else {"halo": what, "ok": False, "error_rank%d" % rank: state["err"][:300]})
            if ok:
                return body, verts, tets, pp, nz, el, host, attempts, halo
            if rank == 0:
                print("[bench] N-rank run failed with: %s -- %s" % (what, state["err"] or "an error on another rank"), file=sys.stderr)
            if body is not None:
                try:
                    body.close()
                except Exception:  # noqa: BLE001
                    pass
        raise SystemExit("the N-rank run failed with every halo setting: " + json.dumps(attempts))
    finally:
        for k in keys:
            os.environ.pop(k, None) if saved[k] is None else os.environ.__setitem__(k, saved[k])

def pmc_traffic(kname, kernel_sha):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json) -- only if they were taken on
    THIS kernel build (same kernel_sha); a stale figure is reported as null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if t.get("kernel_sha") != kernel_sha:
            return None
        for k, v in t.items():   # (template kernels are listed as "void name<0>" or "void name<(int)0>")
            if isinstance(v, dict) and k.replace("void ", "").replace("(int)", "") == kname:
                return v.get("hbm_bytes_per_launch")
        return None
    except Exception:
        return None

def rocprof_kernel_stats(kname, kernel_sha):
    """The committed `rocprofv3 --kernel-trace --stats` summary of this very command (profiles/bench_kernel_stats.json names the CSV of
    the round and the kernel_sha it was taken on): average duration of `kname` over EVERY launch of the command, profiler attached.
    The line carries it beside its own event timings so that the two cannot drift apart unnoticed; a summary taken on another kernel
    build is reported as stale."""
    try:
        with open(os.path.join(ROOT, "profiles", "bench_kernel_stats.json")) as f:
            meta = json.load(f)
        import csv
        with open(os.path.join(ROOT, "profiles", meta["csv"])) as f:
            for row in csv.DictReader(f):
                name = row["Name"].replace("tetsim::(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("(int)", "").strip()
                if name == kname:   # ("void tetsim::(anonymous namespace)::k<0>(...)" -> "k<0>")
                    return {"file": "profiles/" + meta["csv"], "launches": int(row["Calls"]), "kernel_us": round(float(row["AverageNs"]) / 1e3, 2),
                            "stale": meta.get("kernel_sha") != kernel_sha}
    except Exception:
        pass
    return None


def tet_kernel_ceiling(kernel_sha):
    """profiles/tet_kernel_ceiling.json (tools/tet_kernel_ceiling.py, ablation build + counters): what bounds pjb_tet_kernel on this
    workload -- its memory floor (the kernel with the rotation iterations switched off) and its vector-issue floor (VALU
    wave-instructions per SIMD x the issue rate this chip sustains)."""
    try:
        with open(os.path.join(ROOT, "profiles", "tet_kernel_ceiling.json")) as f:
            c = json.load(f)
        return dict(c, stale=c.get("kernel_sha") != kernel_sha)
    except Exception:
        return None


def run(args, rank, world, local_rank, ranks):
    """One rank of the benchmark.  `ranks` is None (single process, no communicator) or an adapter with broadcast_bytes /
    barrier / max_float / min_float."""
    use_dist = ranks is not None
    from tetsim_amd import library_info, measure_copy_bandwidth

    cells = args.cells
    if args.solver == "neohookean":
        if world > 1:
            raise SystemExit("--solver neohookean is a single-GPU benchmark: Gauss-Seidel would need one halo per colour (replicas only)")
        from tetsim_amd import make_lattice
        verts, tets = make_lattice(cells)
        out, body = run_neohookean(args, verts, tets, local_rank)
        out["library"] = library_info()
        return out, body
    attempts = []
    used_halo = None
    if use_dist and world > 1:
        body, verts, tets, pp, nz, elapsed_local, host_local, attempts, used_halo = headline_with_retries(args, cells, rank, world, local_rank, ranks)
    else:
        body, verts, tets, pp, nz, _ = make_body(args, cells, args.scaling, rank, world, local_rank, ranks)
        # ---- timed region --------------------------------------------------------------------------------
        elapsed_local, host_local = timed_frames(body, pp, args.steps, args.warmup, ranks)
        if not np.isfinite(body.pos).all():
            raise SystemExit("non-finite positions after the timed region")
    nt_global = len(tets)
    # tetsim_step_n of this body is ONE launch per call (pj_blocked.hip: pjb_call_kernel, TetSimInfo.fused_particle_pass 5): the dominant
    # kernel is then the whole call -- tiles and particles of its 20 substeps -- and is timed as such (events around each launch)
    one_launch = world == 1 and args.solver == "polar" and int(body.info.fused_particle_pass) == 5
    elapsed = ranks.max_float(elapsed_local) if use_dist else elapsed_local
    mg = multi_gpu_report(body, world, elapsed_local, host_local, args.steps, ranks) if use_dist else None
    if mg is not None and len(attempts) > 1:
        mg["halo_attempts"] = attempts   # (the ones before the last failed: the headline was measured with the last one's settings)

    lib = library_info()
    out = None
    if rank == 0:
        value = nt_global * SUBSTEPS * args.steps / elapsed / 1e6
        nv_global = len(verts)
        out = {
            "metric": "tet_solves_per_sec", "value": round(value, 1), "unit": "M tet-solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Kuhn-6 cube lattice %dx%dx%d cells (%d tets, %d particles), polar-decomposition Jacobi, "
                                   "%d substeps/frame, dt=1/1200 s" % (cells, cells, nz, nt_global, nv_global, SUBSTEPS),
                       "solver": "polar_jacobi", "arithmetic": args.precision,
                       "formulation": "constant rest shape (opt-in)" if args.constant_rest_shape else "lean state (opt-in: three carried corners, no quaternion, 92 B/tet)" if args.lean_state else "reference (rest shape carried from substep to substep, 148 B/tet)", "substeps_per_step": SUBSTEPS,
                       "rotation_exit": "|omega| < 1e-9 throughout (--reference-rotation-exit: the reference's, SoftbodyGPU.js:131)" if getattr(args, "reference_rotation_exit", False) else ("iteration 1: |omega| < 1e-9 (the reference's, SoftbodyGPU.js:131); correction iterations 2..9: |omega| < 1e-6 rad "
                                         "(FAST default; value_reference_threshold at the top level has the same frames with 1e-9 throughout, and roofline.frac is quoted on THAT kernel)") if args.precision == "fast"
                                        else "|omega| < 1e-9 (the reference's, SoftbodyGPU.js:131)",
                       "tets": nt_global, "particles": nv_global,
                       "parallelism": "single GPU" if world == 1 else "z-slab domain decomposition x%d, RCCL ghost halo per substep" % world},
            "library": lib,
        }
        if world == 1:
            out["host_enqueue_us_per_substep"] = round(host_local / (args.steps * SUBSTEPS) * 1e6, 2)
        if mg is not None:
            out["multi_gpu"] = mg
            if used_halo in ("p2p", "deep"):
                out["config"]["parallelism"] = "z-slab domain decomposition x%d, peer-to-peer ghost halo per substep%s (RCCL for set-up, the refresh after a dt change and validation)" % (
                    world, " over a two-layer ghost region" if used_halo == "deep" else "")
    # dominant kernel: its OWN begin/end HIP events (hipExtLaunchKernelGGL) on the handle's stream, inside the real
    # tet -> particle -> tet ... sequence, 60 substeps right after the timed region (same kernels as the graph).  N > 1: every
    # rank takes part (the substeps exchange halos as usual); rank 0 reports ITS interior tet kernel -- the boundary tiles run
    # beside it on the halo stream.
    pr = None
    replay = None
    if world == 1 and args.solver == "polar" and not args.no_replay:
        # The contract's window for the dominant kernel is the TIMED REGION.  Its launches there are graph nodes (no events of their own),
        # so the same W + K frames are stepped once more on a second body with the library's per-launch events (tetsim_profile: begin /
        # end events around every kernel, on the handle's stream, same kernels in the same order).  The trajectory is deterministic:
        # the second body must end bit-equal to the first, and the line says whether it did.
        try:
            pos_timed = body.pos
            body2, _, _, pp2, _, _ = make_body(args, cells, args.scaling, rank, world, local_rank, None)
            for _ in range(args.warmup):
                body2.profile(SUBSTEPS, DT, pp2)
            acc = {"tet_ms": 0.0, "tet_launches": 0, "vertex_ms": 0.0, "vertex_launches": 0}
            for _ in range(args.steps):
                p = body2.profile(SUBSTEPS, DT, pp2)
                for k in acc:
                    acc[k] += p[k]
                acc["tets_per_tet_launch"] = p["tets_per_tet_launch"]
            replay = dict(acc, bit_equal=bool(np.array_equal(body2.pos, pos_timed)))
            body2.close()
        except Exception as e:  # noqa: BLE001  (the line then reports the window after the timed region, as rounds 1-3 did)
            print("[bench] the replay of the timed frames failed: %r" % (e,), file=sys.stderr)
            replay = None
    # EQUAL WORK: the FAST default ends a tet's correction iterations below 1e-6 rad, which removes ~40% of the
    # iterations while the body falls and none once it lies on the floor.  The same W + K frames once more with the REFERENCE's
    # threshold (TETSIM_FLAG_REF_ROTATION_EXIT: |omega| < 1e-9, all nine iterations in f32, SoftbodyGPU.js:131): wall clock (median of
    # three bodies from rest) -> value_reference_threshold, and per-launch events on a fourth -> the kernel the roofline fraction leads with.
    equal = None
    if world == 1 and args.solver == "polar" and args.precision == "fast" and not args.no_replay and rank == 0:
        try:
            from tetsim_amd import SoftBodyHIP
            kw = dict(constant_rest_shape=True) if args.constant_rest_shape else dict(lean_state=True) if args.lean_state else {}
            runs = []
            for _ in range(3):
                b3 = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision="fast", device=local_rank, ref_rotation_exit=True, **kw)
                el3, _ = timed_frames(b3, pp, args.steps, args.warmup, None)
                runs.append(el3)
                b3.close()
            b4 = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision="fast", device=local_rank, ref_rotation_exit=True, **kw)
            for _ in range(args.warmup):
                b4.profile(SUBSTEPS, DT, pp)
            acc4 = {"tet_ms": 0.0, "tet_launches": 0}
            for _ in range(args.steps):
                p4 = b4.profile(SUBSTEPS, DT, pp)
                for k in acc4:
                    acc4[k] += p4[k]
            b4.close()
            equal = {"elapsed": sorted(runs)[1], "runs": runs, "tet_us": acc4["tet_ms"] / acc4["tet_launches"] * 1e3, "launches": acc4["tet_launches"]}
        except Exception as e:  # noqa: BLE001
            print("[bench] the equal-work (reference threshold) leg failed: %r" % (e,), file=sys.stderr)
    # THE CALL AS ONE LAUNCH: its own begin / end HIP events on the handle's stream around every frame's launch (tetsim_time_step_n), on
    # bodies of their own stepped exactly like the headline body -- with the reference's rotation threshold (equal work: what `frac` is
    # quoted on) and with the FAST exit (what `value` ran) --, over the K timed frames and, ~45 frames in, on the floor.
    def call_windows(**kw):
        from tetsim_amd import SoftBodyHIP
        b = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision="fast", device=local_rank, **kw)
        for _ in range(args.warmup):
            b.simulateSubsteps(SUBSTEPS, DT, pp)
        b.sync()
        timed = [b.timeSubsteps(SUBSTEPS, DT, pp) for _ in range(args.steps)]
        for _ in range(max(0, 45 - args.steps - args.warmup)):
            b.simulateSubsteps(SUBSTEPS, DT, pp)
        floor = sorted(b.timeSubsteps(SUBSTEPS, DT, pp) for _ in range(9))
        fin = bool(np.isfinite(b.pos).all())
        b.close()
        return {"timed_ms": sum(timed) / len(timed), "floor_ms": floor[4], "launches": len(timed), "finite": fin}
    callk = None
    if one_launch and args.precision == "fast" and not args.no_replay and rank == 0:
        try:
            kwf = dict(constant_rest_shape=True) if args.constant_rest_shape else dict(lean_state=True) if args.lean_state else {}
            callk = {"ref": call_windows(ref_rotation_exit=True, **kwf), "fast": call_windows(**kwf)}
        except Exception as e:  # noqa: BLE001
            print("[bench] the call-kernel windows failed: %r" % (e,), file=sys.stderr)
    # THE LEAN TET RECORD (TETSIM_FLAG_LEAN_STATE): the same frames on bodies that stream 92 instead of 148 B per
    # tet (three carried corners, no quaternion) -- wall clock with the FAST exit (value_lean) and with the reference's threshold
    # (value_lean_reference_threshold), medians of three bodies from rest each, and the kernel on the floor by per-launch events + in graphs.
    lean = None
    if world == 1 and args.solver == "polar" and args.precision == "fast" and not args.no_replay and not args.no_lean and rank == 0 and not args.constant_rest_shape and not args.lean_state:
        try:
            from tetsim_amd import SoftBodyHIP

            def lean_runs(**kw):
                runs = []
                for _ in range(3):
                    b = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision="fast", device=local_rank, lean_state=True, **kw)
                    el, _ = timed_frames(b, pp, args.steps, args.warmup, None)
                    runs.append(el)
                    b.close()
                return runs
            fast_runs, ref_runs = lean_runs(), lean_runs(ref_rotation_exit=True)
            lean_call = {"ref": call_windows(lean_state=True, ref_rotation_exit=True), "fast": call_windows(lean_state=True)} if one_launch else None
            b5 = SoftBodyHIP(verts, tets, None, dict(pp), solver="polar", precision="fast", device=local_rank, lean_state=True, ref_rotation_exit=True)
            for _ in range(args.warmup):
                b5.profile(SUBSTEPS, DT, pp)
            acc5 = {"tet_ms": 0.0, "tet_launches": 0, "vertex_ms": 0.0, "vertex_launches": 0}
            for _ in range(args.steps):
                p5 = b5.profile(SUBSTEPS, DT, pp)
                for k in acc5:
                    acc5[k] += p5[k]
            # ... on to the floor (the headline body lies there after its W + K frames and three batches; this one needs the same ~45 frames)
            for _ in range(max(0, 45 - args.steps - args.warmup)):
                b5.simulateSubsteps(SUBSTEPS, DT, pp)
            fl = sorted((b5.profile(SUBSTEPS * 3, DT, pp) for _ in range(3)), key=lambda p: p["tet_ms"] / p["tet_launches"])[1]
            b5.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                b5.simulateSubsteps(SUBSTEPS, DT, pp)
            b5.sync()
            floor_sub = (time.perf_counter() - t0) / (10 * SUBSTEPS) * 1e6
            finite = bool(np.isfinite(b5.pos).all())
            b5.close()
            lean = {"fast_runs": fast_runs, "ref_runs": ref_runs, "timed_tet_us": acc5["tet_ms"] / acc5["tet_launches"] * 1e3, "timed_launches": acc5["tet_launches"],
                    "floor_tet_us": fl["tet_ms"] / fl["tet_launches"] * 1e3, "floor_vert_us": fl["vertex_ms"] / max(fl["vertex_launches"], 1) * 1e3,
                    "floor_substep_us": floor_sub, "finite": finite, "call": lean_call}
        except Exception as e:  # noqa: BLE001
            print("[bench] the lean-state leg failed: %r" % (e,), file=sys.stderr)
    if world == 1 or (args.profile_ranks and args.precision == "fast"):
        # three batches of 60 substeps, the median batch is reported (a single batch right after the timed region is
        # occasionally 5-8% slow on a box that agrees with rocprofv3 otherwise)
        batches = sorted((body.profile(SUBSTEPS * 3, DT, pp) for _ in range(3)), key=lambda p: p["tet_ms"] / p["tet_launches"])
        pr = batches[1]
        body.sync()
        floor_graph_us = None
        if world == 1 and args.solver == "polar":
            # ... and what a substep costs there INSIDE the graphs (the regime of the timed region: kernels back to back, no per-launch events)
            try:
                t0 = time.perf_counter()
                for _ in range(10):
                    body.simulateSubsteps(SUBSTEPS, DT, pp)
                body.sync()
                floor_graph_us = (time.perf_counter() - t0) / (10 * SUBSTEPS) * 1e6
            except Exception as e:  # noqa: BLE001
                print("[bench] the on-floor graph frames failed: %r" % (e,), file=sys.stderr)
        if use_dist:
            ranks.barrier()
    tet_bytes = TET_KERNEL_BYTES_LEAN if args.lean_state else TET_KERNEL_BYTES - (48.0 if args.constant_rest_shape else 0.0)   # constant rest shape: read only, never written back
    b_alg = tet_bytes + VERTEX_BYTES * len(verts) / len(tets)
    if rank == 0 and pr is not None:
        after_us = pr["tet_ms"] / pr["tet_launches"] * 1e3
        win = replay if replay is not None and replay["tet_launches"] else pr
        tet_us = win["tet_ms"] / win["tet_launches"] * 1e3
        vert_us = win["vertex_ms"] / win["vertex_launches"] * 1e3 if win["vertex_launches"] else 0.0
        units = win["tets_per_tet_launch"]
        kname = ("pjb_tet_kernel_lean" if args.lean_state else "pjb_tet_kernel") if args.precision == "fast" else "pj_tet_kernel_precise"
        if body.info.fused_particle_pass in (1, 2):   # small bodies (< 2,048 tiles): one kernel per substep does the particle row too
            kname, tet_bytes = "pjb_tet_kernel_x<TetFused>", b_alg
        traffic = pmc_traffic(kname, lib["kernel_sha"]) if world == 1 and not args.constant_rest_shape and not args.lean_state and cells == CELLS else None
        alg = tet_bytes * units

        def window(us, what, **more):
            return dict({"kernel_us": round(us, 2), "achieved": round(alg / (us * 1e-6) / 1e9, 1), "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                         "window": what}, **more)
        # Which duration `frac` is quoted on.  EQUAL WORK FIRST: nine rotation iterations in every tet, as the reference does
        # (SoftbodyGPU.js:122-139).  The line LEADS with the window after the timed frames: the body lies on the floor, every tet runs
        # nine iterations whatever the threshold, the clocks have settled -- the window rounds 1-4 can be compared on.  Beside it:
        # the timed frames with the reference's threshold (equal work while the body falls; by events, and what the wall clock of those
        # graphs implies -- the two disagree on the first frames after a body's creation) and the FAST-exit kernel `value` ran.
        fast_win = window(tet_us, ("the %d timed frames (%d launches) of the headline body's trajectory, stepped again on a second body with per-launch "
                                   "events; trajectory bit-equal to the timed one: %s" % (args.steps, replay["tet_launches"], replay["bit_equal"]))
                          if win is replay else "180 substeps after the timed region, median of three batches of 60",
                          rotation_exit="FAST default: correction iterations end below 1e-6 rad (~40% fewer iterations while the body falls)")
        floor_win = window(after_us, "180 substeps after the timed region (the body lies on the floor: nine iterations in every tet whatever the threshold), "
                                     "per-launch events, median of three batches of 60")
        lead = floor_win
        # (one launch per call: tiles and particles of neighbouring substeps overlap, "substep minus particle kernel minus boundaries" means nothing)
        boundaries_us = (elapsed / (args.steps * SUBSTEPS) * 1e6 - tet_us - vert_us) if (win is replay and world == 1 and not one_launch) else None
        if floor_graph_us is not None and boundaries_us is not None:
            floor_win["in_graph"] = {"substep_us": round(floor_graph_us, 2), "kernel_us_implied": round(floor_graph_us - vert_us - boundaries_us, 2),
                                     "frac_implied": round(alg / ((floor_graph_us - vert_us - boundaries_us) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                     "what": "10 more frames as tetsim_step_n calls (wall clock) minus the particle kernel and the two launch boundaries of timed_region_check"}
        ref_win = None
        if equal is not None:
            ref_win = window(equal["tet_us"], "the %d timed frames (%d launches) with the reference's rotation threshold (|omega| < 1e-9), per-launch events on a "
                                              "body of its own" % (args.steps, equal["launches"]))
            sub_ref = equal["elapsed"] / (args.steps * SUBSTEPS) * 1e6
            if boundaries_us is not None:
                imp = sub_ref - vert_us - boundaries_us
                ref_win["in_graph"] = {"substep_us": round(sub_ref, 2), "kernel_us_implied": round(imp, 2), "frac_implied": round(alg / (imp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                       "what": "wall clock of the same frames as tetsim_step_n calls (value_reference_threshold) minus the particle kernel and the two launch "
                                               "boundaries of timed_region_check; on the first frames after a body's creation the graphs run the iteration-heavy kernel "
                                               "slower than its per-launch events say (tools/attic/frame_series.py)"}
        # WHAT THE PRODUCT RUNS LEADS: the equal-work kernel INSIDE the graphs tetsim_step_n replays -- the timed
        # frames with the reference's threshold: their wall clock per substep minus the particle kernel and the two launch boundaries --
        # is `frac`; the per-launch event figures (about 1 us shorter: an eagerly launched kernel with its own events starts on an idle chip)
        # stand beside it as frac_events / on_floor / timed_frames_reference_threshold.
        events_lead = lead
        if ref_win is not None and "in_graph" in ref_win:
            ig = ref_win["in_graph"]
            lead = {"kernel_us": ig["kernel_us_implied"], "achieved": round(alg / (ig["kernel_us_implied"] * 1e-6) / 1e9, 1), "frac": ig["frac_implied"],
                    "window": "the %d timed frames with the reference's rotation threshold AS GRAPH REPLAYS (what tetsim_step_n runs): wall clock per substep (%.2f us, "
                              "median of three bodies) minus the particle kernel (%.2f us) and the two launch boundaries (%.2f us) of timed_region_check"
                              % (args.steps, ig["substep_us"], vert_us, boundaries_us)}
        elif "in_graph" in floor_win:
            ig = floor_win["in_graph"]
            lead = {"kernel_us": ig["kernel_us_implied"], "achieved": round(alg / (ig["kernel_us_implied"] * 1e-6) / 1e9, 1), "frac": ig["frac_implied"],
                    "window": "ten frames on the floor as tetsim_step_n calls: " + ig["what"]}
        achieved = lead["achieved"]
        out["roofline"] = {"bound": "hbm", "kernel": kname + ("" if world == 1 else " (rank 0, interior tiles)"),
                           "achieved": lead["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": lead["frac"], "traffic": traffic,
                           "kernel_us": lead["kernel_us"], "window": lead["window"],
                           "frac_events": events_lead["frac"], "kernel_us_events": events_lead["kernel_us"], "window_events": events_lead["window"],
                           "work": "equal to the reference's: nine rotation iterations in every tet (SoftbodyGPU.js:122-139)" if args.precision == "fast" else "the reference's",
                           "vertex_kernel_us": round(vert_us, 2),
                           "alg_bytes_per_launch": alg,
                           "substep_alg_bytes_per_tet": round(b_alg, 1),
                           "substep_achieved": round(b_alg * out["value"] * 1e6 / 1e9, 1),
                           "substep_frac": round(b_alg * out["value"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)}
        if args.precision == "fast" and world == 1:
            out["roofline"]["on_floor"] = floor_win
            if ref_win is not None:
                out["roofline"]["timed_frames_reference_threshold"] = ref_win
            out["roofline"]["fast_exit"] = fast_win
            out["roofline"]["after_timed_region"] = floor_win       # (the name rounds 1-4 used)
            rp = rocprof_kernel_stats(kname, lib["kernel_sha"])
            if rp is not None:
                rp["frac"] = round(alg / (rp["kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                rp["what"] = ("rocprofv3 --kernel-trace --stats over this very command, profiler attached: the average over EVERY launch of the kernel "
                              "(headline frames with the FAST exit, the replays, the reference-threshold bodies, the on-floor window)")
                out["roofline"]["frac_rocprof"] = rp["frac"]
                out["roofline"]["rocprof"] = rp
            ce = tet_kernel_ceiling(lib["kernel_sha"])
            if ce is not None:
                ceil_us = max(ce["memory_floor_us"], ce["valu_issue_floor_us"])
                out["roofline"]["ceiling"] = dict(ce, ceiling_us=ceil_us, frac_at_ceiling=round(alg / (ceil_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                  kernel_vs_ceiling=round(ceil_us / events_lead["kernel_us"], 4), kernel_vs_ceiling_in_graph=round(ceil_us / lead["kernel_us"], 4))   # (the floors are event-timed: like against like first)
        if equal is not None:
            out["value_reference_threshold"] = round(nt_global * SUBSTEPS * args.steps / equal["elapsed"] / 1e6, 1)
            out["value_reference_threshold_runs"] = [round(nt_global * SUBSTEPS * args.steps / r / 1e6, 1) for r in equal["runs"]]
            # the whole substep at equal work, against the peak: SURVEY 8(d)'s 173.3 B per tet-solve x Nt over the in-graph substep of those frames
            out["roofline"]["frac_substep_reference_threshold"] = round(b_alg * out["value_reference_threshold"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)
        if lean is not None:
            rate = lambda el: round(nt_global * SUBSTEPS * args.steps / el / 1e6, 1)   # noqa: E731
            b_alg_lean = TET_KERNEL_BYTES_LEAN + VERTEX_BYTES * len(verts) / len(tets)
            alg_lean = TET_KERNEL_BYTES_LEAN * units
            out["value_lean"] = rate(sorted(lean["fast_runs"])[1])
            out["value_lean_runs"] = [rate(r) for r in lean["fast_runs"]]
            out["value_lean_reference_threshold"] = rate(sorted(lean["ref_runs"])[1])
            out["value_lean_reference_threshold_runs"] = [rate(r) for r in lean["ref_runs"]]
            sub_ref_lean = sorted(lean["ref_runs"])[1] / (args.steps * SUBSTEPS) * 1e6
            rl = {"bound": "hbm", "kernel": "pjb_tet_kernel_lean", "formulation": "TETSIM_FLAG_LEAN_STATE: three carried corners in and out, no quaternion in the substep "
                  "(recovered from the carried shape at read-out): positions 16 + shape 36 + 36 + weight 4 = 92 algorithmic B per tet -- ITS OWN B_alg (SURVEY.md 8(d)), not the reference formulation's 148",
                  "alg_bytes_per_tet": TET_KERNEL_BYTES_LEAN, "alg_bytes_per_launch": alg_lean, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "substep_alg_bytes_per_tet": round(b_alg_lean, 1), "finite": lean["finite"]}
            if boundaries_us is not None:   # what leads: the equal-work kernel inside the graphs (as for the reference formulation above)
                imp = sub_ref_lean - lean["floor_vert_us"] - boundaries_us
                rl.update({"kernel_us": round(imp, 2), "achieved": round(alg_lean / (imp * 1e-6) / 1e9, 1), "frac": round(alg_lean / (imp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "window": "the %d timed frames with the reference's rotation threshold as tetsim_step_n calls: wall clock per substep (%.2f us, median of three bodies) minus the "
                                     "particle kernel (%.2f us) and the two launch boundaries (%.2f us)" % (args.steps, sub_ref_lean, lean["floor_vert_us"], boundaries_us)})
            else:
                rl.update({"kernel_us": round(lean["floor_tet_us"], 2), "achieved": round(alg_lean / (lean["floor_tet_us"] * 1e-6) / 1e9, 1),
                           "frac": round(alg_lean / (lean["floor_tet_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "window": "on the floor, per-launch events"})
            ev = lambda us, what: {"kernel_us": round(us, 2), "achieved": round(alg_lean / (us * 1e-6) / 1e9, 1), "frac": round(alg_lean / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "window": what}   # noqa: E731
            rl["on_floor"] = dict(ev(lean["floor_tet_us"], "180 substeps on the floor (nine iterations in every tet), per-launch events, median of three batches of 60"),
                                  in_graph={"substep_us": round(lean["floor_substep_us"], 2), "what": "ten more frames as tetsim_step_n calls, wall clock per substep"})
            rl["timed_frames_reference_threshold"] = ev(lean["timed_tet_us"], "the %d timed frames (%d launches) with the reference's rotation threshold, per-launch events" % (args.steps, lean["timed_launches"]))
            rl["vertex_kernel_us"] = round(lean["floor_vert_us"], 2)
            rl["substep_frac_reference_threshold"] = round(b_alg_lean * out["value_lean_reference_threshold"] * 1e6 / 1e9 / HBM_PEAK_GBS, 4)
            ce = tet_kernel_ceiling(lib["kernel_sha"])
            if ce is not None and ce.get("lean"):
                # the leaner record is no longer bound by bytes: its ceiling is the vector-issue floor of its own instruction count
                lc = ce["lean"]
                ceil_us = max(lc["memory_floor_us"], lc["valu_issue_floor_us"])
                rl["ceiling"] = dict(lc, ceiling_us=ceil_us, kernel_vs_ceiling=round(ceil_us / rl["kernel_us"], 4), stale=ce["stale"])
            rl["speedup_vs_reference_formulation"] = {"value": round(out["value_lean"] / out["value"], 4),
                                                      "value_reference_threshold": round(out["value_lean_reference_threshold"] / out["value_reference_threshold"], 4) if "value_reference_threshold" in out else None}
            out["roofline_lean"] = rl
        # the FAST-exit replay against the wall clock of the timed region itself: a substep there is the two kernels plus two launch boundaries
        if win is replay and world == 1 and not one_launch:
            sub_us = elapsed / (args.steps * SUBSTEPS) * 1e6
            out["roofline"]["timed_region_check"] = {"substep_us": round(sub_us, 2), "kernels_us": round(tet_us + vert_us, 2),
                                                     "two_launch_boundaries_us": round(sub_us - tet_us - vert_us, 2),
                                                     "what": "the headline's own frames (FAST exit): wall clock per substep against the replayed kernels"}
        if world == 1:
            # SURVEY.md 8(d) "bounding roofline": the peak is also MEASURED on this box -- a device copy at the footprint class of
            # the 1 M-tet working set (fits the 256 MB Infinity Cache) and at 1 GiB (streams from HBM)
            # (the tuned probe of tetsim_measure_stream_bandwidth: four independent 16-byte accesses per lane, plain / non-temporal and the
            # grid size chosen at first use; read-only and write-only rates beside the copy rate)
            from tetsim_amd import measure_stream_bandwidth
            cp = {"64MiB": round(measure_copy_bandwidth(64 << 20, 20), 0), "1GiB": round(measure_copy_bandwidth(1 << 30, 10), 0)}
            out["roofline"]["measured_copy_peak"] = cp
            out["roofline"]["measured_stream_peak_1GiB"] = {"read": round(measure_stream_bandwidth(1 << 30, "read", 10), 0),
                                                            "write": round(measure_stream_bandwidth(1 << 30, "write", 10), 0), "unit": "GB/s"}
            out["roofline"]["frac_of_measured_peak"] = {"kernel_vs_64MiB_copy": round(achieved / cp["64MiB"], 4),
                                                        "kernel_vs_1GiB_copy": round(achieved / cp["1GiB"], 4),
                                                        "substep_vs_64MiB_copy": round(b_alg * out["value"] * 1e6 / 1e9 / cp["64MiB"], 4),
                                                        "substep_vs_1GiB_copy": round(b_alg * out["value"] * 1e6 / 1e9 / cp["1GiB"], 4)}
    if rank == 0 and pr is not None and one_launch and callk is not None:
        # The dominant kernel of this body IS the call: pjb_call_kernel, one launch per frame (per substep every tile, then every particle).
        # Its algorithmic bytes per launch are the WHOLE substep's (tet row + particle row = b_alg per tet-solve) x tets x 20;
        # its duration comes from its own events around every launch.  What the kernel pair of rounds 1-5 measured (tetsim_step / tetsim_profile
        # still run it, same bits) stays below as two_kernel_path.
        tk = out["roofline"]
        kcall = {"carried": "pjb_call_kernel<0>", "lean": "pjb_call_kernel<2>", "constant": "pjb_call_kernel<1>"}["lean" if args.lean_state else "constant" if args.constant_rest_shape else "carried"]
        alg_call = b_alg * nt_global * SUBSTEPS

        def cw(ms, what, alg=alg_call):
            us = ms * 1e3
            return {"kernel_us": round(us, 1), "substep_us": round(us / SUBSTEPS, 2), "achieved": round(alg / (us * 1e-6) / 1e9, 1),
                    "frac": round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "window": what}
        how = "; one launch per frame of %d substeps, begin / end HIP events on the handle's stream around each launch (tetsim_time_step_n)" % SUBSTEPS
        lead = cw(callk["ref"]["timed_ms"], "the %d timed frames with the reference's rotation threshold (|omega| < 1e-9: nine iterations in every tet) on a body of its own" % args.steps + how)
        rf = {"bound": "hbm", "kernel": kcall + ": the tiles and the particles of a call's %d substeps in ONE launch, handed on by stamped partial sums / predictions" % SUBSTEPS,
              "achieved": lead["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": lead["frac"], "traffic": pmc_traffic(kcall, lib["kernel_sha"]) if cells == CELLS else None,
              "kernel_us": lead["kernel_us"], "substep_us": lead["substep_us"], "window": lead["window"],
              "work": "equal to the reference's: nine rotation iterations in every tet (SoftbodyGPU.js:122-139)",
              "alg_bytes_per_launch": alg_call, "alg_bytes_per_tet_solve": round(b_alg, 1), "launches": callk["ref"]["launches"],
              "on_floor": cw(callk["ref"]["floor_ms"], "nine launches ~45 frames in (the body lies on the floor), median" + how),
              "fast_exit": cw(callk["fast"]["timed_ms"], "the %d timed frames with the FAST exit (what `value` ran: correction iterations end below 1e-6 rad)" % args.steps + how),
              "fast_exit_on_floor": cw(callk["fast"]["floor_ms"], "nine launches ~45 frames in, FAST exit, median" + how),
              "headline_wall_clock": {"substep_us": round(elapsed / (args.steps * SUBSTEPS) * 1e6, 2),
                                      "frac": round(alg_call / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                                      "what": "the headline's own timed region (FAST exit, no events inside): wall clock per frame against the same bytes"},
              "substep_alg_bytes_per_tet": round(b_alg, 1), "substep_achieved": tk.get("substep_achieved"), "substep_frac": tk.get("substep_frac")}
        if "value_reference_threshold" in out:
            rf["frac_substep_reference_threshold"] = round(b_alg * out["value_reference_threshold"] * 1e6 / 1e9 / HBM_PEAK_GBS, 4)
        rp = rocprof_kernel_stats(kcall, lib["kernel_sha"])
        if rp is not None:
            rp["frac"] = round(alg_call / (rp["kernel_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            rp["what"] = "rocprofv3 --kernel-trace --stats over this very command, profiler attached: the average over EVERY launch of the kernel (FAST exit and reference threshold, falling and on the floor)"
            rf["frac_rocprof"], rf["rocprof"] = rp["frac"], rp
        for k in ("measured_copy_peak", "measured_stream_peak_1GiB"):
            if k in tk:
                rf[k] = tk[k]
        if "measured_copy_peak" in tk:
            cp = tk["measured_copy_peak"]
            rf["frac_of_measured_peak"] = {"kernel_vs_64MiB_copy": round(lead["achieved"] / cp["64MiB"], 4), "kernel_vs_1GiB_copy": round(lead["achieved"] / cp["1GiB"], 4)}
        tk["what"] = ("the same substeps as a tet kernel + a particle kernel (tetsim_step, tetsim_profile: per-launch events; rounds 1-5's dominant kernel pjb_tet_kernel "
                      "against its own 148 B per tet) -- same results bit for bit")
        rf["two_kernel_path"] = tk
        out["roofline"] = rf
        rl = out.get("roofline_lean")
        if rl is not None and lean is not None and lean.get("call"):
            b_alg_lean = TET_KERNEL_BYTES_LEAN + VERTEX_BYTES * len(verts) / len(tets)
            alg_lc = b_alg_lean * nt_global * SUBSTEPS
            lc = lean["call"]
            rl["two_kernel_path"] = {k: rl.pop(k) for k in ("kernel", "kernel_us", "achieved", "frac", "window", "on_floor", "timed_frames_reference_threshold", "vertex_kernel_us", "alg_bytes_per_launch") if k in rl}
            top = cw(lc["ref"]["timed_ms"], "the %d timed frames with the reference's rotation threshold on a lean-state body of its own" % args.steps + how, alg_lc)
            rl.update({"kernel": "pjb_call_kernel<2> (lean state)", "kernel_us": top["kernel_us"], "substep_us": top["substep_us"], "achieved": top["achieved"], "frac": top["frac"], "window": top["window"],
                       "alg_bytes_per_launch": alg_lc, "alg_bytes_per_tet_solve": round(b_alg_lean, 1),
                       "on_floor": cw(lc["ref"]["floor_ms"], "nine launches ~45 frames in, median" + how, alg_lc),
                       "fast_exit": cw(lc["fast"]["timed_ms"], "the %d timed frames with the FAST exit (what value_lean ran)" % args.steps + how, alg_lc),
                       "fast_exit_on_floor": cw(lc["fast"]["floor_ms"], "nine launches ~45 frames in, FAST exit, median" + how, alg_lc)})

    def whole_job_roofline():
        # N > 1 without --profile-ranks: the whole-job figure only (no extra GPU work after the timed region)
        agg = b_alg * out["value"] * 1e6 / 1e9
        return {"bound": "hbm", "kernel": "whole substep, all ranks (tet + particle kernels)", "achieved": round(agg, 1),
                "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": round(agg / (HBM_PEAK_GBS * world), 4), "traffic": None,
                "substep_alg_bytes_per_tet": round(b_alg, 1)}
    if rank == 0 and pr is None:
        out["roofline"] = whole_job_roofline()
    # ---- optional legs of an N-rank run: nothing below may cost the headline (HeadlineGuard) ------------------------------------
    if use_dist and world > 1:
        GUARD.arm(out, int(os.environ.get("TETSIM_BENCH_OPTIONAL_S", "240")), "the legs after the headline (peer-to-peer halo check / config 5)")
        hp = halo_probe_report(body, ranks)   # (the body is between steps: the timed region ended with a sync + barrier)
        if rank == 0 and out is not None and "multi_gpu" in out:
            out["multi_gpu"]["halo_exchange_us"] = hp
    check = args.p2p_check == "on" or (args.p2p_check == "auto" and not args.fake_ranks)
    if use_dist and world > 1 and used_halo in ("p2p", "deep") and check:
        # The headline ran on the peer-to-peer halo.  RCCL is the VALIDATOR: the same frames from the same rest state on a fresh body whose
        # ghosts travel by grouped ncclSend / ncclRecv must end bit-equal (one-layer ghost regions; the two-layer region has another
        # rounding path and is compared for finiteness only) -- and its rate stands beside the headline (multi_gpu.rccl_halo).  A run
        # that is NOT confirmed does not keep the headline: the RCCL figures take it (demote_p2p).
        try:
            pos_head = body.pos
        except Exception:  # noqa: BLE001
            pos_head = None
        res = transport_check(args, cells, rank, world, local_rank, ranks, pos_head, nt_global, "rccl")
        if rank == 0:
            out["multi_gpu"]["rccl_halo"] = res
            out["multi_gpu"]["headline_validated_against_rccl"] = bool(isinstance(res, dict) and not res.get("error") and res.get("finite") and (res.get("bit_equal_to_headline_run") or used_halo == "deep"))
            if demote_p2p(out, res, args.steps, world, exact=used_halo == "p2p"):
                if pr is None:
                    out["roofline"] = whole_job_roofline()
                else:
                    out["roofline"]["substep_achieved"] = round(b_alg * out["value"] * 1e6 / 1e9, 1)
                    out["roofline"]["substep_frac"] = round(b_alg * out["value"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)
    elif use_dist and world > 1 and used_halo == "rccl" and args.halo == "rccl" and check:
        # --halo rccl: RCCL first (rounds 1-5's order), the peer-to-peer halo afterwards as a validated second run that may take the headline
        try:
            pos_rccl = body.pos
        except Exception:  # noqa: BLE001
            pos_rccl = None
        res = p2p_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global)
        if rank == 0:
            out["multi_gpu"]["p2p_halo"] = res
            if promote_p2p(out, res, args.steps, world, args.headline_halo):
                if pr is None:
                    out["roofline"] = whole_job_roofline()
                else:
                    out["roofline"]["substep_achieved"] = round(b_alg * out["value"] * 1e6 / 1e9, 1)
                    out["roofline"]["substep_frac"] = round(b_alg * out["value"] * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)
    # ---- BASELINE config 5, literally: the 110^3-cell lattice (7,986,000 tets) cut into N slabs -- strong scaling -----------
    if use_dist and (args.config5 == "on" or (args.config5 == "auto" and world == 8)) and not (args.scaling == "strong" and cells == args.config5_cells):
        # (the headline body stays alive: if this second body cannot be built on some rank, the line above is still reported)
        body5, v5, t5, pp5, _, err5 = make_body(args, args.config5_cells, "strong", rank, world, local_rank, ranks, vote=True, halo=used_halo)   # (the transport the headline ended up with)
        if body5 is None:
            if rank == 0:
                out["config5_strong"] = {"error": err5}
        else:
            body.close()
            body = body5
            steps5 = max(1, min(args.steps, 10))
            e5_local, h5_local = timed_frames(body, pp5, steps5, min(args.warmup, 2), ranks)
            e5 = ranks.max_float(e5_local)
            finite = ranks.min_float(1.0 if np.isfinite(body.pos).all() else 0.0)
            mg5 = multi_gpu_report(body, world, e5_local, h5_local, steps5, ranks)
            if rank == 0:
                v = len(t5) * SUBSTEPS * steps5 / e5 / 1e6
                out["config5_strong"] = {
                    "workload": "Kuhn-6 cube lattice %d^3 cells (%d tets, %d particles) cut into %d z-slabs, %d-particle interface planes, "
                                "polar-decomposition Jacobi, %d substeps/frame" % (args.config5_cells, len(t5), len(v5), world, (args.config5_cells + 1) ** 2, SUBSTEPS),
                    "scaling": "strong", "value": round(v, 1), "unit": "M tet-solves/s", "steps": steps5, "ms_per_step": round(e5 / steps5 * 1e3, 4),
                    "finite": bool(finite), "multi_gpu": mg5,
                    "substep_frac_of_hbm_roofline": round(b_alg * v * 1e6 / 1e9 / (HBM_PEAK_GBS * world), 4)}
    GUARD.disarm()
    if world == 1:
        body.close()

        def side_leg(fn):   # (nothing behind the headline and its roofline may cost the line: a failing side leg reports itself)
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                print("[bench] side leg failed: %r" % (e,), file=sys.stderr)
                return {"error": repr(e)[:300]}
        if not args.no_beyond_mall and args.precision == "fast" and cells == CELLS and args.solver == "polar" and "roofline" in out:
            out["roofline"]["beyond_mall"] = side_leg(lambda: beyond_mall(args, local_rank, out["roofline"].get("measured_copy_peak", {})))
        if not args.no_other_configs and args.precision == "fast" and cells == CELLS:
            out["other_configs"] = side_leg(lambda: other_configs(args.steps, args.warmup))
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = side_leg(lambda: cpu_baseline(verts, tets))
    return out, body
