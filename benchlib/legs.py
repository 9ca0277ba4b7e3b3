"""Everything the bench line carries BESIDE the headline, all outside the timed region: BASELINE configs 1 / 2 / 4 and the reference
rotation threshold (other_configs), config 4 on request (run_neohookean), the beyond-Infinity-Cache body (beyond_mall), the N-rank
run's self-diagnosis (multi_gpu_report) and the validated peer-to-peer halo run (p2p_check, promote_p2p)."""
import json
import os
import sys
import time

import numpy as np

from .body import make_body
from .common import CELLS, DT, GOLD, HBM_PEAK_GBS, PP, ROOT, SUBSTEPS, TET_KERNEL_BYTES, VERTEX_BYTES

def other_configs(steps=20, warmup=5):
    """BASELINE configs 1, 2 and 4 on this box, outside the timed region, a few seconds in total.  (Config 3 is the line itself,
    config 5 needs N > 1.)  Same metric everywhere: M tet-solves/s = tets x substeps / wall."""
    import shutil
    import subprocess
    from tetsim_amd import SoftBodyHIP, make_lattice
    out = {}
    dv = np.fromfile(os.path.join(GOLD, "dragon_verts.f32"), dtype="<f4").reshape(-1, 3)
    dtets = np.fromfile(os.path.join(GOLD, "dragon_tets.i32"), dtype="<i4").reshape(-1, 4)

    def hip_rate(v, t, n_sub, frames, **kw):
        body = SoftBodyHIP(v, t, None, dict(PP), **kw)
        dt = (PP["timeScale"] * PP["timeStep"]) / n_sub
        body.simulateSubsteps(n_sub, dt, PP)
        body.sync()
        t0 = time.perf_counter()
        for _ in range(frames):
            body.simulateSubsteps(n_sub, dt, PP)
        body.sync()
        el = time.perf_counter() - t0
        levels = body.info.num_levels
        mode = int(body.info.fused_particle_pass)   # 0: tet + particle kernel per substep; 1: one fused kernel per substep; 2 / 3: one persistent kernel per frame (3: four lanes per tet)
        body.close()
        return {"value": round(len(t) * n_sub * frames / el / 1e6, 2), "unit": "M tet-solves/s", "ms_per_frame": round(el / frames * 1e3, 4),
                "us_per_substep": round(el / frames / n_sub * 1e6, 2), "frames": frames,
                "launches_per_substep": round(1.0 / n_sub, 3) if mode == 4 else (levels + 1) if levels else {0: 2, 1: 1, 2: round(1.0 / n_sub, 3), 3: round(1.0 / n_sub, 3), 5: round(1.0 / n_sub, 3)}[mode],
                "kernel": {0: "tet + particle kernel per substep", 1: "fused kernel per substep", 2: "persistent frame kernel, one lane per tet (256-tet tiles)",
                           3: "persistent frame kernel, four lanes per tet (64-tet tiles)", 5: "the call's tiles and particles in one launch (stamped hand-overs)"}[mode] if not levels
                          else "Gauss-Seidel levels, one launch per frame: one workgroup, every particle in LDS" if mode == 4 else "Gauss-Seidel levels, one launch per level"}

    # config 1: Dragon, the reference's CPU solver (Neo-Hookean Gauss-Seidel), 10 substeps per frame
    c1 = {"workload": "Dragon (%d tets, %d particles), Neo-Hookean XPBD Gauss-Seidel, 10 substeps/frame" % (len(dtets), len(dv))}
    node = shutil.which("node")
    if node:
        try:
            r = subprocess.run([node, os.path.join(ROOT, "oracle", "nh_port.js"), "--verts", os.path.join(GOLD, "dragon_verts.f32"), "--tets",
                                os.path.join(GOLD, "dragon_tets.i32"), "--substeps", "400", "--warmup", "100", "--per-frame", "10"],
                               capture_output=True, text=True, timeout=120)
            jr = json.loads(r.stdout)
            c1["softbody_js_algorithm_node_1thread"] = {"value": round(jr["m_tet_solves_per_s"], 3), "unit": "M tet-solves/s", "cores": 1, "kind": "port",
                                                        "sample": "400 substeps after 100 warm-up, oracle/nh_port.js under node " + jr["node"]}
        except Exception as e:  # node is optional
            c1["softbody_js_algorithm_node_1thread"] = {"error": str(e)[:200]}
    c1["hip_original_order_precise"] = dict(hip_rate(dv, dtets, 10, 10, solver="neohookean", precision="precise", order="original"),
                                            note="bit-exact with Softbody.js in the caller's tet order (703 dependency levels)")
    c1["hip_coloured_precise"] = dict(hip_rate(dv, dtets, 10, 100, solver="neohookean", precision="precise", order="coloured"),
                                      note="bit-exact with Softbody.js fed tetIds[tetsim_get_tet_order()]")
    c1["hip_coloured_fast"] = dict(hip_rate(dv, dtets, 10, 100, solver="neohookean", precision="fast", order="coloured"),
                                   note="f32 arithmetic, tolerance contract (tests/golden/tolerances.json)")
    out["config1_dragon_neohookean_cpu_path"] = c1
    # config 2: Dragon, polar-decomposition Jacobi, f32, 20 substeps per frame
    out["config2_dragon_polar_jacobi"] = {
        "workload": "Dragon, polar-decomposition Jacobi, 20 substeps/frame, one graph launch per frame (FAST: ONE persistent kernel per frame, "
                    "every tile's workgroup resident for the 20 substeps; PRECISE: a tet and a particle kernel per substep)",
        "fast": hip_rate(dv, dtets, 20, 400, solver="polar", precision="fast"),
        "precise": hip_rate(dv, dtets, 20, 200, solver="polar", precision="precise")}
    # (config 3 with the REFERENCE's rotation-exit threshold is at the top level of the line: value_reference_threshold, headline.py)
    # config 4: Neo-Hookean Gauss-Seidel on the 1 M-tet lattice + convergence against Jacobi (dropped 2 cm onto the floor)
    v, t = make_lattice(CELLS, y0=0.02)
    Dm_inv = np.linalg.inv((v[t[:, 1:]] - v[t[:, :1]]).astype(np.float64).transpose(0, 2, 1))

    def vol_residual(pos):   # mean |det F - 1|: the reference's volError analogue (Softbody.js:163)
        F = (pos[t[:, 1:]] - pos[t[:, :1]]).astype(np.float64).transpose(0, 2, 1) @ Dm_inv
        return float(np.abs(np.linalg.det(F) - 1.0).mean())

    c4 = {"workload": "Kuhn-6 lattice %d^3 cells (%d tets) dropped 2 cm onto the floor, %d substeps/frame" % (CELLS, len(t), SUBSTEPS),
          "residual": "mean |det F - 1| after 1 / 5 / 30 frames, evaluated on the host in f64 from the returned positions"}
    for key, kw in (("neohookean_clustered_gs_fast", dict(solver="neohookean", precision="fast", order="clustered")),
                    ("neohookean_clustered_gs_precise", dict(solver="neohookean", precision="precise", order="clustered")),
                    ("polar_jacobi_fast", dict(solver="polar", precision="fast"))):
        body = SoftBodyHIP(v, t, None, dict(PP), **kw)
        snaps, done = [], 0
        for frames in (1, 5, 30):
            for _ in range(frames - done):
                body.simulateSubsteps(SUBSTEPS, DT, PP)
            done = frames
            snaps.append(body.pos.copy())   # (the f64 residual of 1 M tets takes the host ~0.3 s: evaluated AFTER the timed frames, so that they do not start from an idle device)
        body.sync()
        t0 = time.perf_counter()   # the rate: 20 more frames of the same body (resting on the floor by now), graph already built
        for _ in range(20):
            body.simulateSubsteps(SUBSTEPS, DT, PP)
        body.sync()
        el = time.perf_counter() - t0
        res = [float("%.3e" % vol_residual(p)) for p in snaps]
        c4[key] = {"value": round(len(t) * SUBSTEPS * 20 / el / 1e6, 1), "unit": "M tet-solves/s", "ms_per_frame": round(el / 20 * 1e3, 4),
                   "mean_abs_detF_minus_1_after_1_5_30_frames": res,
                   "launches_per_substep": (body.info.num_levels + (0 if body.info.fused_particle_pass else 1)) if body.info.num_levels else 2}
        body.close()
    out["config4_lattice_1m_neohookean_gs_vs_jacobi"] = c4
    return out

def run_neohookean(args, verts, tets, device):
    """BASELINE config 4 on request (`--solver neohookean`): Neo-Hookean XPBD Gauss-Seidel (Softbody.js's algorithm, coloured or
    clustered schedule) on the same lattice and metric.  PRECISE reproduces Softbody.js bit for bit on the permuted tet order."""
    from tetsim_amd import SoftBodyHIP
    body = SoftBodyHIP(verts, tets, None, dict(PP), solver="neohookean", precision=args.precision, order=args.order, device=device)
    for _ in range(args.warmup):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    elapsed = time.perf_counter() - t0
    if not np.isfinite(body.pos).all():
        raise SystemExit("non-finite positions after the timed region")
    value = len(tets) * SUBSTEPS * args.steps / elapsed / 1e6
    b_alg = 56.0 + 124.0 * len(verts) / len(tets)   # idx 16 + invRestPose 36 + invRestVolume 4; 124 B per particle
    agg = b_alg * value * 1e6 / 1e9
    pr = body.profile(SUBSTEPS * 3, DT, PP)
    out = {
        "metric": "tet_solves_per_sec", "value": round(value, 1), "unit": "M tet-solves/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if args.precision == "precise" else "f32", "data": "synthetic",
        "config": {"workload": "Kuhn-6 cube lattice %dx%dx%d cells (%d tets, %d particles), Neo-Hookean XPBD Gauss-Seidel (%s schedule, "
                               "%d launches per substep), %d substeps/frame, dt=1/1200 s" % (args.cells, args.cells, args.cells, len(tets), len(verts), args.order,
                                                                                           body.info.num_levels, SUBSTEPS),
                   "solver": "neohookean_gs", "arithmetic": args.precision, "order": args.order, "substeps_per_step": SUBSTEPS,
                   "tets": len(tets), "particles": len(verts), "parallelism": "single GPU"},
        # the bound of this solver is its dependency chain (launches x (launch + round trips) + sequential tet solves);
        # the HBM figure is reported because the contract asks for one
        "roofline": {"bound": "hbm", "kernel": "whole substep (Gauss-Seidel sweep + particle pass)", "achieved": round(agg, 1),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(agg / HBM_PEAK_GBS, 4), "traffic": None,
                     "sweep_us_per_substep": round(pr["tet_ms"] / pr["substeps"] * 1e3, 2),
                     "particle_us_per_substep": round(pr["vertex_ms"] / pr["substeps"] * 1e3, 2),
                     "substep_alg_bytes_per_tet": round(b_alg, 1)},
    }
    if not args.no_cpu_baseline:
        body.close()
        from oracle import OracleNH
        nh = OracleNH(verts, tets, PP)
        nh.simulate(DT, PP)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 10.0:
            nh.simulate(DT, PP)
            n += 1
        out["cpu_baseline"] = {"value": round(n * len(tets) / (time.perf_counter() - t0) / 1e6, 3), "unit": "M tet-solves/s", "cores": 1,
                               "kind": "port", "sample": "%d substeps of the same lattice, sequential Gauss-Seidel in the caller's tet order "
                                                         "(oracle/tetsim_oracle.c section A: Softbody.js's algorithm, bit-exact with its goldens)" % n}
    return out, body

def beyond_mall(args, device, copy_peak, cells=110, frames=10):
    """SURVEY.md 8(d) asks for a figure at a size beyond the 256 MB Infinity Cache as well: the 110^3-cell lattice (7,986,000 tets,
    ~1.3 GB of per-tet state) on this one GPU, same kernels, `frames` frames after 2 warm-up frames, then the dominant kernel's own
    events over 20 substeps.  Outside the timed region of the headline; ~3 s incl. building the body."""
    from tetsim_amd import SoftBodyHIP, make_lattice
    t_build = time.perf_counter()
    v, t = make_lattice(cells)
    kw = {"constant_rest_shape": True} if args.constant_rest_shape else {}
    body = SoftBodyHIP(v, t, None, dict(PP), solver="polar", precision="fast", device=device, **kw)
    t_build = time.perf_counter() - t_build
    for _ in range(2):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    t0 = time.perf_counter()
    for _ in range(frames):
        body.simulateSubsteps(SUBSTEPS, DT, PP)
    body.sync()
    el = time.perf_counter() - t0
    finite = bool(np.isfinite(body.pos).all())
    pr = body.profile(SUBSTEPS, DT, PP)
    body.close()
    value = len(t) * SUBSTEPS * frames / el / 1e6
    tet_bytes = TET_KERNEL_BYTES - (48.0 if args.constant_rest_shape else 0.0)
    b_alg = tet_bytes + VERTEX_BYTES * len(v) / len(t)
    tet_us = pr["tet_ms"] / pr["tet_launches"] * 1e3
    ach = tet_bytes * pr["tets_per_tet_launch"] / (tet_us * 1e-6) / 1e9
    res = {"workload": "Kuhn-6 cube lattice %d^3 cells (%d tets, %d particles), same solver and kernels, %d frames of %d substeps" % (cells, len(t), len(v), frames, SUBSTEPS),
           "value": round(value, 1), "unit": "M tet-solves/s", "ms_per_step": round(el / frames * 1e3, 4), "finite": finite,
           "kernel_us": round(tet_us, 2), "vertex_kernel_us": round(pr["vertex_ms"] / pr["vertex_launches"] * 1e3, 2) if pr["vertex_launches"] else 0.0,
           "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
           "substep_achieved": round(b_alg * value * 1e6 / 1e9, 1), "substep_frac": round(b_alg * value * 1e6 / 1e9 / HBM_PEAK_GBS, 4),
           "build_s": round(t_build, 2)}
    if copy_peak.get("1GiB"):
        res["frac_of_1GiB_copy"] = round(ach / copy_peak["1GiB"], 4)
        res["substep_frac_of_1GiB_copy"] = round(b_alg * value * 1e6 / 1e9 / copy_peak["1GiB"], 4)
    return res

def multi_gpu_report(body, world, elapsed_local, host_local, steps, ranks):
    """What makes an N-rank run self-diagnosing: RCCL's own rank count (must equal --gpus), the spread of the ranks' step times,
    this rank's halo volume, the host enqueue time per substep."""
    from tetsim_amd import comm_info
    ci = comm_info(body)
    if ci["rccl_ranks"] != world:
        raise SystemExit("RCCL reports %d ranks in the halo communicator but --gpus is %d: refusing to report a number" % (ci["rccl_ranks"], world))
    ms = elapsed_local / steps * 1e3
    hq = host_local / (steps * SUBSTEPS) * 1e6
    rep = {"rccl_ranks": ci["rccl_ranks"], "halo": ("p2p" if ci.get("p2p") else "rccl") + (" (two-layer ghost region, ghosts every other substep)" if body.info.flags & 32 else ""),
           "ranks_ms_per_step": {"min": round(ranks.min_float(ms), 4), "max": round(ranks.max_float(ms), 4)},
           "host_enqueue_us_per_substep": {"min": round(ranks.min_float(hq), 2), "max": round(ranks.max_float(hq), 2)},
           "halo_rank0": {"neighbours": ci["neighbours"], "send_bytes_per_substep": ci["send_bytes_per_substep"],
                          "recv_bytes_per_substep": ci["recv_bytes_per_substep"], "max_message_bytes": ci["max_message_bytes"]},
           "halo_max_message_bytes_over_ranks": int(ranks.max_float(float(ci["max_message_bytes"]))),
           "owned_tets_rank0": int(body.info.owned_elems), "local_tets_rank0": int(body.info.local_elems)}
    if ci["loopback"]:
        rep["loopback"] = True
    return rep

def halo_probe_report(body, ranks):
    """The transfer term of the substep's halo chain on THIS wire: 100 exchanges of the real messages with the real neighbours (a
    collective of the ranks; idempotent).  The chain is  wait V + halo-side tiles + boundary particles (~19 us on one GPU)  +  this.
    Runs under the HeadlineGuard: nothing here may cost the line."""
    def one(probe, what):
        # (a local failure is VOTED on before the ranks' reductions: nobody is left alone inside a collective)
        hp, err = None, None
        try:
            hp = probe(body, 100)
        except Exception as e:  # noqa: BLE001
            err = repr(e)[:200]
        if ranks.min_float(0.0 if err else 1.0) < 1.0:
            return {"error": err or "the probe failed on another rank"}
        return {"rank0": {k: round(v, 1) for k, v in hp.items()}, "median_min_over_ranks": round(ranks.min_float(hp["median"]), 1),
                "median_max_over_ranks": round(ranks.max_float(hp["median"]), 1), "what": what}
    from tetsim_amd import comm_info, halo_p2p_probe, halo_probe
    rep = one(halo_probe, "grouped ncclSend / ncclRecv of this rank's halo messages to its neighbours, events on the halo stream, 100 repetitions between steps")
    if comm_info(body).get("p2p") and not (body.info.flags & 32):
        # the run used the peer-to-peer halo: its hand-over on this wire beside RCCL's exchange of the same messages
        rep = dict(rep, rccl=dict(rep), p2p=one(halo_p2p_probe, "one store into every neighbour's inbox word + one wait on the own ones (tetsim_halo_p2p_probe): the one-way signal "
                                                                "latency of the peer-to-peer halo on this wire, device clock, 100 repetitions between steps"))
    return rep


def p2p_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global):
    """--halo rccl: the RCCL headline run once more with the peer-to-peer halo (transport_check); `bit_equal_to_rccl_run` is the name
    rounds 3-5 used for the comparison."""
    res = transport_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global, "p2p")
    if isinstance(res, dict) and "bit_equal_to_headline_run" in res:
        res["bit_equal_to_rccl_run"] = res["bit_equal_to_headline_run"]
    return res


def transport_check(args, cells, rank, world, local_rank, ranks, pos_rccl, nt_global, halo):
    """The headline run once more on a fresh body whose halo goes over the OTHER transport (`halo`: "rccl" after a peer-to-peer headline,
    "p2p" after an RCCL one; include/tetsim.h: tetsim_halo_p2p_connect): the same warm-up and timed frames from the same rest state, so
    the owned positions must equal the headline run's BIT FOR BIT -- on real peers, which the one-GPU tests cannot show -- and the rate
    says what RCCL's send/recv kernel on the substep's chain costs here.  Every local step is caught and VOTED on (a rank never leaves
    the others inside a collective), device-side waits are short, one probe substep comes first, and the whole leg sits under the
    HeadlineGuard's budget."""
    saved = os.environ.get("TETSIM_HALO_TIMEOUT_MS")
    os.environ["TETSIM_HALO_TIMEOUT_MS"] = "4000"
    state = {"err": None}

    def local(fn):      # run a local step unless this rank has failed already; remember the first failure
        if state["err"] is None:
            try:
                return fn()
            except Exception as e:  # noqa: BLE001
                state["err"] = "rank %d: %r" % (rank, e)
        return None

    def everyone_ok():  # collective
        return ranks.min_float(0.0 if state["err"] else 1.0) >= 1.0

    body2 = None
    try:
        body2, _, _, pp2, _, err = make_body(args, cells, args.scaling, rank, world, local_rank, ranks, vote=True, halo=halo)
        if body2 is None:
            return {"error": err}
        local(lambda: (body2.simulate(DT, pp2), body2.sync()))   # a transport that does not work shows here, within seconds
        if not everyone_ok():
            return {"error": state["err"] or "the probe substep failed on another rank"}
        local(lambda: body2.simulateSubsteps(SUBSTEPS - 1, DT, pp2))
        # the first frame is done; the others as in the headline run: the rest of the warm-up untimed, then the timed frames
        frames_before = max(args.warmup - 1, 0)
        timed = args.warmup + args.steps - 1 - frames_before
        for _ in range(frames_before):
            local(lambda: body2.simulateSubsteps(SUBSTEPS, DT, pp2))
        local(body2.sync)
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(timed):
            local(lambda: body2.simulateSubsteps(SUBSTEPS, DT, pp2))
        local(body2.sync)
        ranks.barrier()
        el_local = time.perf_counter() - t0
        pos2 = local(lambda: body2.pos)
        same = pos2 is not None and pos_rccl is not None and bool(np.array_equal(pos2.view(np.uint32), pos_rccl.view(np.uint32)))
        fin = pos2 is not None and bool(np.isfinite(pos2).all())
        el = ranks.max_float(el_local)
        res = {"value": round(nt_global * SUBSTEPS * timed / el / 1e6, 1) if timed > 0 else None, "unit": "M tet-solves/s",
               "ms_per_step": round(el / max(timed, 1) * 1e3, 4), "steps": timed,
               "halo": halo, "bit_equal_to_headline_run": bool(ranks.min_float(1.0 if same else 0.0) >= 1.0), "finite": bool(ranks.min_float(1.0 if fin else 0.0) >= 1.0),
               "ranks_ms_per_step": {"min": round(ranks.min_float(el_local / max(timed, 1) * 1e3), 4), "max": round(ranks.max_float(el_local / max(timed, 1) * 1e3), 4)}}
        if not everyone_ok():
            res["error"] = state["err"] or "a step failed on another rank"
        return res
    finally:
        if body2 is not None:
            try:
                ranks.barrier()
                body2.close()
            except Exception:  # noqa: BLE001
                pass
        if saved is None:
            os.environ.pop("TETSIM_HALO_TIMEOUT_MS", None)
        else:
            os.environ["TETSIM_HALO_TIMEOUT_MS"] = saved

def demote_p2p(out, res, steps, world, exact=True):
    """A peer-to-peer headline that its RCCL validator run (`res`, transport_check) does NOT confirm -- positions not bit-equal (exact:
    one-layer ghost regions), or non-finite -- gives the headline to the RCCL run's figures, measured over the same frames under the same
    protocol; the unconfirmed figures stay in multi_gpu.p2p_halo_unconfirmed.  A validator that could not run at all (error) leaves the
    headline where it is, flagged (multi_gpu.headline_validated_against_rccl: false).  Returns whether `out` was changed."""
    if not isinstance(res, dict) or res.get("error") or not res.get("value") or res.get("steps") != steps:
        return False
    if res.get("finite") and (res.get("bit_equal_to_headline_run") or not exact):
        return False
    mgr = out["multi_gpu"]
    mgr["p2p_halo_unconfirmed"] = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "ranks_ms_per_step": mgr.get("ranks_ms_per_step")}
    out["value"], out["ms_per_step"] = res["value"], res["ms_per_step"]
    mgr["ranks_ms_per_step"] = res.get("ranks_ms_per_step")
    mgr["halo"] = "rccl: the peer-to-peer run of the same frames was NOT confirmed by this RCCL run (positions differ or are not finite): the RCCL figures are the headline"
    out["config"]["parallelism"] = "z-slab domain decomposition x%d, RCCL ghost halo per substep" % world
    return True


def promote_p2p(out, res, steps, world, mode="best"):
    """The headline of an N-rank run is the faster of the two halo transports -- if the peer-to-peer run (`res`, p2p_check) is VALIDATED
    in this very run: the same frames from the same rest state under the same protocol (barrier, synchronise, max over ranks), every
    rank's positions equal to the RCCL run's bit for bit.  Otherwise, or with mode "rccl", the RCCL figures in `out` stand.  Returns
    whether `out` was changed (value, ms_per_step, multi_gpu.halo / ranks_ms_per_step / rccl_halo, config.parallelism)."""
    if mode != "best" or not isinstance(res, dict) or res.get("error") or not res.get("bit_equal_to_rccl_run") or not res.get("finite"):
        return False
    if res.get("steps") != steps or not res.get("value") or res["value"] <= out["value"]:
        return False
    mgr = out["multi_gpu"]
    mgr["rccl_halo"] = {"value": out["value"], "unit": out["unit"], "ms_per_step": out["ms_per_step"], "ranks_ms_per_step": mgr.get("ranks_ms_per_step")}
    out["value"], out["ms_per_step"] = res["value"], res["ms_per_step"]
    mgr["ranks_ms_per_step"] = res.get("ranks_ms_per_step")
    mgr["halo"] = ("p2p: boundary particles stored straight into the neighbours' IPC-mapped ghost ranges -- the faster of the two transports, validated in "
                   "this run (positions bit-equal to the RCCL run of the same frames, whose figures are in multi_gpu.rccl_halo)")
    out["config"]["parallelism"] = "z-slab domain decomposition x%d, peer-to-peer ghost halo per substep (RCCL for set-up and validation)" % world
    return True
