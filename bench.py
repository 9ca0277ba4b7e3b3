#!/usr/bin/env python3
"""bench.py -- headline benchmark: M tet-solves/s on the 1 M-tet Kuhn lattice, polar-decomposition Jacobi,
20 substeps per frame, on N MI355X (BASELINE.json metric; SURVEY.md §8(d) config 3, config 5 shape for N>1).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 without a launcher: spawns its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one animation frame of the reference's driver loop (main.js:79-84): 20 substeps of the hot path
over the whole body, issued as ONE tetsim_step_n call (one HIP-graph launch).  All state is resident in HBM
before the timed region; nothing is read back inside it.  N > 1: weak scaling -- the lattice is stacked to
55 x 55 x 55N cells, slab-partitioned along z (one slab, ~1 M tets, per GPU), ghost-vertex positions cross
xGMI once per substep through RCCL send/recv issued by libtetsim_hip itself.

The JSON line carries two extra objects:
  roofline      the dominant kernel (pj_tet_kernel, P3+P4 of the reference) against HBM peak.  `achieved` =
                algorithmic bytes per launch / mean launch duration measured here with HIP events on the
                handle's own stream.  Bytes per tet are SURVEY.md §8(d)'s tet row (148 B) -- see DESIGN.md.
  cpu_baseline  the CPU restatement (oracle/, "port") of the same algorithm on this host's cores, bounded sample.
and, outside the timed region:
  library        what was loaded: ABI, source / kernel hashes, "ablation": false for the product build, debug env knobs set
  other_configs  (N = 1) BASELINE configs 1, 2 and 4 on the same box, bounded to a few seconds
  multi_gpu      (N > 1) ranks RCCL itself reports, per-rank step time min/max, halo bytes, host enqueue time per substep;
                 the run FAILS if RCCL's rank count differs from --gpus
  config5_strong (N = 8, or --config5 on) BASELINE config 5 literally: the 110^3-cell lattice cut into N slabs

This file is the entry point (arguments, launch modes); the parts live in benchlib/: launcher.py (self-launch, one-line stdout, the
headline guard), ranks.py (torch.distributed / thread-rank adapters), body.py + headline.py (the body, the timed region, the retry
ladder, the JSON line and its roofline object), legs.py (everything beside the headline, after the timed region), cpu.py (the CPU
baseline: the only part, with legs.py's node / Gauss-Seidel CPU legs, that touches oracle/).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.common import CELLS  # noqa: E402
from benchlib.cpu import cpu_baseline, cpu_budget  # noqa: E402,F401
from benchlib.body import make_body, timed_frames  # noqa: E402,F401
from benchlib.headline import headline_with_retries, run  # noqa: E402,F401
from benchlib.launcher import GUARD, HeadlineGuard, self_launch, stdout_to_stderr  # noqa: E402,F401
from benchlib.legs import other_configs, p2p_check, promote_p2p  # noqa: E402,F401
from benchlib.ranks import ThreadRanks, TorchRanks  # noqa: E402


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 200 frames = 4000 substeps = 0.17 s timed: past the clock ramp of a short run
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="fast", choices=["fast", "precise"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--solver", default="polar", choices=["polar", "neohookean"],
                    help="polar (default: the headline, BASELINE configs 3/5) or neohookean (config 4: coloured Gauss-Seidel on the same "
                         "lattice; single GPU only -- it does not partition)")
    ap.add_argument("--order", default="clustered", choices=["coloured", "clustered"], help="--solver neohookean: Gauss-Seidel schedule")
    ap.add_argument("--constant-rest-shape", action="store_true",
                    help="opt-in TETSIM_FLAG_CONSTANT_REST_SHAPE formulation (NOT the headline: 100 instead of 148 algorithmic B/tet)")
    ap.add_argument("--lean-state", action="store_true",
                    help="the HEADLINE body with TETSIM_FLAG_LEAN_STATE (92 instead of 148 algorithmic B/tet: three carried corners, no quaternion in the substep) -- "
                         "counter passes and profiles of the lean kernel; the default line already carries value_lean / roofline_lean beside the reference formulation")
    ap.add_argument("--no-lean", action="store_true", help="N = 1: skip the lean-state leg (value_lean, roofline_lean)")
    ap.add_argument("--reference-rotation-exit", action="store_true",
                    help="the HEADLINE body with TETSIM_FLAG_REF_ROTATION_EXIT (|omega| < 1e-9: nine rotation iterations in every tet, the reference's "
                         "work) -- counter passes of the equal-work kernel; the default line already carries value_reference_threshold beside value")
    ap.add_argument("--cells", type=int, default=CELLS, help="lattice cells per side (default 55 = the 1 M-tet headline; 110 = 8 M tets)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver measures): cells^2 x (cells*N) lattice, one slab per GPU; strong: the cells^3 "
                         "lattice split into N slabs (BASELINE config 5 is --scaling strong --cells 110)")
    ap.add_argument("--profile-ranks", action="store_true",
                    help="N > 1: after the timed region every rank runs 60 more substeps with per-kernel events and rank 0 reports "
                         "its interior tet kernel in `roofline` (default at N > 1: whole-substep figures only)")
    ap.add_argument("--config5", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: append BASELINE config 5 literally (the --config5-cells^3 lattice cut into N slabs, strong scaling) as "
                         "`config5_strong` of the same JSON line; auto = when N == 8")
    ap.add_argument("--config5-cells", type=int, default=110, help="cells per side of the config-5 body (110 = 7,986,000 tets)")
    ap.add_argument("--no-replay", action="store_true", help="N = 1: time the dominant kernel in the 180 substeps AFTER the timed region only (rounds 1-3), not over a replay of the timed frames")
    ap.add_argument("--no-beyond-mall", action="store_true", help="N = 1: skip `roofline.beyond_mall` (the 8 M-tet body on this GPU)")
    ap.add_argument("--no-other-configs", action="store_true", help="N = 1: skip the `other_configs` object (BASELINE configs 1, 2, 4)")
    ap.add_argument("--halo", default="p2p", choices=["p2p", "rccl", "deep"],
                    help="N > 1: transport of the per-substep ghost exchange the HEADLINE run starts with -- p2p (default: the boundary-particle kernel stores "
                         "straight into the neighbours' ghost ranges, mapped through HIP IPC; an RCCL run of the same frames afterwards validates it bit for "
                         "bit and is reported beside it; a failed vote falls back to RCCL), rccl (RCCL first: grouped ncclSend/ncclRecv, the peer-to-peer "
                         "halo as a validated second run that may take the headline -- rounds 1-5's order) or deep (p2p over a two-layer ghost region: "
                         "ghosts cross every other substep, TETSIM_FLAG_DEEP_GHOSTS)")
    ap.add_argument("--p2p-check", default="auto", choices=["auto", "on", "off"],
                    help="N > 1: the second run over the OTHER transport on a fresh body -- after a peer-to-peer headline the RCCL validator "
                         "(`multi_gpu.rccl_halo`, `multi_gpu.headline_validated_against_rccl`), with --halo rccl the peer-to-peer run (`multi_gpu.p2p_halo`); "
                         "auto = on, except with --fake-ranks")
    ap.add_argument("--headline-halo", default="best", choices=["best", "rccl"],
                    help="--halo rccl only: best (default) = report the faster transport as the headline IF the peer-to-peer run was validated in this run "
                         "(bit-equal positions, same frames, same protocol), with the RCCL figures beside it in multi_gpu.rccl_halo; rccl = the RCCL run is the "
                         "headline whatever the check says")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank code path even with one rank (smoke test)")
    ap.add_argument("--self-spawn", action="store_true",
                    help="launch the rank processes from this process even when --gpus is 1 (what a plain `python bench.py --gpus N`, "
                         "N > 1, does by itself when no launcher set WORLD_SIZE)")
    ap.add_argument("--fake-ranks", type=int, default=0,
                    help="development: run N ranks as threads on ONE GPU against the RCCL test double (needs TETSIM_RCCL_LIB=tests/mock_rccl/...)")
    return ap.parse_args()


def main():
    # a rank that is stuck in a collective would otherwise hang the whole launch until someone's outer limit fires
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("TETSIM_BENCH_WATCHDOG_S", "600")), exit=True)
    args = parse_args()
    if args.fake_ranks > 1:
        import threading
        if "mock_rccl" not in os.environ.get("TETSIM_RCCL_LIB", ""):
            raise SystemExit("--fake-ranks needs TETSIM_RCCL_LIB to point at the test double (tests/mock_rccl/libmock_rccl.so)")
        os.environ["TETSIM_HALO_GRAPH"] = "0"   # the test double rendezvouses on the host: not capturable
        n = args.fake_ranks
        shared = {"barrier": threading.Barrier(n), "vals": [0.0] * n, "bytes": None}
        results, errors = [None] * n, []

        def rank_main(r):
            try:
                out, body = run(args, r, n, 0, ThreadRanks(shared, r))
                shared["barrier"].wait()
                body.close()
                results[r] = out
            except BaseException as e:  # noqa: BLE001
                errors.append("rank %d: %r" % (r, e))
                shared["barrier"].abort()

        with stdout_to_stderr():
            threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(n)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
        if errors:
            raise SystemExit("; ".join(errors))
        print(json.dumps(results[0]), flush=True)
        faulthandler.cancel_dump_traceback_later()
        return

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.self_spawn):
        # a plain `python bench.py --gpus N` (the driver's N = 1 command with N changed): no launcher gave this process a rank, so
        # it becomes the launcher -- one rank process per GPU, rank 0's JSON line forwarded (the torchrun launch keeps working:
        # there WORLD_SIZE is set and this branch is not taken)
        faulthandler.cancel_dump_traceback_later()
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.self_spawn:
        args.force_dist = True      # a spawned rank always takes the torch.distributed path, also when it is the only one
    with stdout_to_stderr():
        ranks = TorchRanks(local_rank, rank, world) if (world > 1 or args.force_dist) else None
        out, body = run(args, rank, world, local_rank, ranks)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if ranks is not None:
        with stdout_to_stderr():
            ranks.barrier()
            body.close()
            ranks.close()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
