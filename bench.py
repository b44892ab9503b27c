#!/usr/bin/env python3
"""Headline benchmark: SPPM iterations per second on cornell_dragon at 1024^2
(200k global + 50k caustic photons per iteration, kNN k = 100 / 50, 1 camera
ray per pixel), synthetic: the scene files under assets/ and random-free
deterministic sampling.

    python bench.py --gpus N --steps K --warmup W

A step is one full SPPM iteration: both photon passes, both KD-tree builds,
the camera pass with its gathers and the per-pixel update.  With N > 1 (one
process per GPU under torch.distributed.run) the fixed workload is split over
the ranks (strong scaling): photon paths and pixel rows are sharded, the
photon maps are all-gathered over RCCL every iteration.

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the program
brings the N ranks up itself: `--driver torch` (default) re-runs itself under
torch.distributed.run as a child process, before anything here has touched a
GPU; `--driver group` keeps one process and drives the N GPUs through the
library's own group entry points (tkhip_group_*: a host thread per GPU, photon
maps pulled with peer copies, no launcher and no torch in the loop), which is
how the reference starts its workers (src/core/TkRayTracer.cpp:198-208).

Prints one JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = "cornell_dragon.scene"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, GB/s


def _oracle_iteration(O, osc, param, max_depth, rr, seed, threads, height, budget_s, **mode):
    """One oracle iteration, the camera pass sampled in 8-row bands spread evenly over the image
    (bit-reversed band order) until budget_s seconds are spent; returns seconds per full iteration."""
    r = O.OracleSppm(osc, param, max_depth, rr, seed, threads=threads, **mode)
    t0 = time.time()
    st_p = r.photon_pass(0)
    t1 = time.time()
    r.build_maps()
    t2 = time.time()
    bands = list(range(0, height, 8))
    bits = max(1, (len(bands) - 1).bit_length())
    order = sorted(range(len(bands)), key=lambda i: int(format(i, f"0{bits}b")[::-1], 2))
    rows = rays = gathers = 0
    for i in order:
        y0 = bands[i]
        y1 = min(y0 + 8, height)
        st = r.camera_pass(0, y0, y1)
        rows += y1 - y0
        rays += st.rays
        gathers += st.gathers_global + st.gathers_caustic
        if time.time() - t0 > budget_s and rows >= 32:
            break
    t3 = time.time()
    tc_photon, tc_camera = r.trace_counters()
    r.close()
    scale = height / rows
    sec = (t2 - t0) + (t3 - t2) * scale
    return {"sec_iter": sec, "photon_s": t1 - t0, "build_s": t2 - t1, "camera_s": t3 - t2, "rows": rows,
            "rays": st_p.rays + rays * scale, "gathers": gathers * scale, "spent_s": t3 - t0,
            "tc_photon": tc_photon, "tc_camera": tc_camera}


def cpu_baseline(scene, width, height, param, max_depth, rr, seed, threads):
    """The CPU oracle (a port of the reference algorithm) on a bounded sample of the same workload,
    in its reference-faithful mode (the reported value) and in a fair mode beside it."""
    from oracle import oracle_py as O
    osc = O.OracleScene(scene, width, height)
    # as the reference computes: z-only KD splits, far-first BVH without t_max culling, one Mersenne
    # Twister construction per random number
    ref = _oracle_iteration(O, osc, param, max_depth, rr, seed, threads, height, 12.0,
                            split_mode=0, faithful_bvh=1, rng_cost=1)
    # the same port with those three costs removed: widest-axis KD-tree, near-first culled BVH, hashed RNG
    fair = _oracle_iteration(O, osc, param, max_depth, rr, seed, threads, height, 8.0,
                             split_mode=1, faithful_bvh=0, rng_cost=0)
    osc.close()

    def describe(m):
        return (f"full photon pass {m['photon_s']:.2f}s + both KD builds {m['build_s']:.2f}s + camera pass on "
                f"{m['rows']} of {height} rows in 8-row bands spread over the image {m['camera_s']:.2f}s "
                f"(scaled x{height / m['rows']:.1f})")
    # Traversal work per ray on the same rays (iteration 0), counted by the port's near-first, culled
    # traversal of the reference's own two-level BVH: what the ray roofline below is priced with.
    per_ray = {}
    for name in ("tc_photon", "tc_camera"):
        rays_, nodes_, prims_ = fair[name]
        per_ray[name] = {"rays": rays_, "box_tests_per_ray": nodes_ / max(rays_, 1), "prim_tests_per_ray": prims_ / max(rays_, 1)}
    # how the port's reference-faithful mode compares with the reference binary itself (measured where the
    # reference builds, oracle/calibrate_cpu_baseline.py; the reference cannot run on the GPU box)
    calibration = None
    cp = os.path.join(ROOT, "profiles", "cpu_baseline_calibration.json")
    if os.path.exists(cp):
        with open(cp) as f:
            cj = json.load(f)
        ratio = float(cj["faithful_over_reference"])
        calibration = {"port_seconds_over_reference_seconds": ratio,
                       "value_reference_estimate": ratio / ref["sec_iter"],
                       "note": "the port takes this share of the reference binary's time per iteration (same scene, size, photons, "
                               "threads; profiles/cpu_baseline_calibration.json): the reference itself would run at about "
                               "value_reference_estimate iterations/s here"}
    return {
        "value": 1.0 / ref["sec_iter"], "unit": "iterations/s", "cores": threads, "kind": "port", "iteration": 0,
        "reference_calibration": calibration,
        "per_ray": per_ray,
        "sample": ("CPU oracle, iteration 0 of the same workload (the GPU's figure for the same iteration is iter0_ms), reference-faithful mode (z-only KD splits, far-first "
                   "BVH without t_max culling, per-call Mersenne-Twister seeding): " + describe(ref)),
        "mrays_per_s": ref["rays"] / ref["sec_iter"] / 1e6, "gathers_per_s": ref["gathers"] / ref["sec_iter"],
        "fair_mode": {"value": 1.0 / fair["sec_iter"], "unit": "iterations/s",
                      "sample": "widest-axis KD-tree, near-first culled BVH, hashed RNG: " + describe(fair),
                      "mrays_per_s": fair["rays"] / fair["sec_iter"] / 1e6,
                      "gathers_per_s": fair["gathers"] / fair["sec_iter"]},
        "cpu_seconds_spent": ref["spent_s"] + fair["spent_s"],
    }


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(gpus, driver, environ, argv, port=None):
    """The command that runs this program as `gpus` ranks, or None when this process is to do the work itself:
    one GPU, a rank that a launcher has already started (WORLD_SIZE set), or the in-process group driver."""
    if gpus <= 1 or "WORLD_SIZE" in environ or driver != "torch":
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default=WORKLOAD)
    ap.add_argument("--size", type=int, default=0, help="override the image size of the scene (square)")
    ap.add_argument("--synthetic", type=int, default=0, metavar="QUADS",
                    help="instead of a scene file: the Cornell shell with two spheres over a height field of QUADS x QUADS "
                         "quads (2236 gives the 10 M-triangle configuration)")
    ap.add_argument("--photons", type=int, nargs=2, default=None, metavar=("GLOBAL", "CAUSTIC"),
                    help="override the photon map sizes of the scene")
    ap.add_argument("--tree", choices=("curve", "median"), default="curve",
                    help="photon tree layout: one sort along a Hilbert curve (default) or median splits level by level")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jobs", type=int, default=40,
                    help="whole renders of the scene's own iteration count timed after the steps (job_ms, iter0_ms); 0 skips them")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--ahead", type=int, default=0, choices=(0, 1, 2),
                    help="N > 1: begin the next iteration's light paths (1) and camera paths (2) under this iteration's gathers; "
                         "a rank alone on a GPU measured 3.50 / 3.71 / 3.99 ms per iteration as rank 0 of 2 and 2.41 / 2.39 / 2.65 as rank 0 of 8")
    ap.add_argument("--driver", choices=("torch", "group"), default="torch",
                    help="N > 1: one process per GPU under torch.distributed.run with an RCCL all-gather per iteration (torch), "
                         "or one process that drives the N GPUs through tkhip_group_* with peer copies (group)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo rehearses N > 1 on fewer GPUs")
    ap.add_argument("--share-gpus", action="store_true",
                    help="rehearsal: ranks take GPU (rank mod GPUs present) instead of one GPU each (implied by --backend gloo)")
    return ap.parse_args(argv)


STAT_KEYS = ("rays", "gathers_global", "gathers_caustic", "gather_photons", "overflow_global", "overflow_caustic",
             "global_paths", "caustic_paths", "group_staged", "rays_camera",
             # of one more iteration run with the counting instances of the path kernels (tkhip_sppm_set_counting)
             "node_visits_camera", "prim_tests_camera", "node_visits_light", "prim_tests_light", "counted_rays", "counted_rays_camera")
COUNT_KEYS = ("node_visits_camera", "prim_tests_camera", "node_visits_light", "prim_tests_light")
TIME_KEYS = ("ms_gather_kernel", "ms_photon", "ms_build", "ms_camera", "ms_gather")


def load_workload(args):
    from pathtracer_amd.scene import Scene, asset_scene
    scene = Scene.synthetic(args.synthetic) if args.synthetic else Scene.load(asset_scene(args.scene))
    name = (f"synthetic height field {args.synthetic}x{args.synthetic} quads ({scene.n_triangles} triangles)"
            if args.synthetic else args.scene)
    W = args.size or scene.width
    H = args.size or scene.height
    param = scene.sppm_param(globalSize=args.photons[0], causticSize=args.photons[1]) if args.photons else scene.sppm_param()
    return scene, name, W, H, param


def totals_of(lib, ctx_h, reset):
    from pathtracer_amd import _abi
    t = _abi.TkGatherTotals()
    _abi.check(lib, lib.tkhip_sppm_gather_totals(ctx_h, C.byref(t), 1 if reset else 0), "tkhip_sppm_gather_totals")
    return t


def add_counted(acc, counted, n_inst):
    """Folds the counters of the counting iteration into a rank's sums (scaled so that sum_stats' division by the
    number of instrumented iterations gives them back)."""
    for k in COUNT_KEYS:
        acc[k] = counted[k] * n_inst
    acc["counted_rays"] = counted["rays"] * n_inst
    acc["counted_rays_camera"] = counted["rays_camera"] * n_inst


def sum_stats(per_rank_dicts, n_inst):
    """Per-iteration averages of the instrumented iterations: counters summed over the ranks, times of the slowest rank."""
    vals = {k: sum(d.get(k, 0) for d in per_rank_dicts) / n_inst for k in STAT_KEYS}
    times = {k: max(d.get(k, 0.0) for d in per_rank_dicts) / n_inst for k in TIME_KEYS}
    return vals, times


def report(args, name, scene, W, H, param, max_depth, rr, seed, world, ranks_seen, driver, elapsed, vals, times, gt,
           job_ms, iter0_ms, n_job_iters):
    """The JSON line.  vals / times: sum_stats of the instrumented iterations; gt: the gather totals of the timed
    region, summed over the ranks (ms_group_kernel: the slowest rank's)."""
    its = args.steps / elapsed
    ms_step = elapsed / args.steps * 1e3
    # Algorithmic bytes of a gather: 56 B query + 26 B per photon found + 40 B pixel slice.
    # The grouped kernel finishes the gathers that met fewer than k photons; the others end
    # with exactly k photons in the per-query kernel and are not charged to it here.
    # Measured over the timed region: every launch of the kernel (one per step: it takes the groups of both
    # maps' gathers) is bracketed by HIP events on the stream it runs on.
    k_g, k_c = param.globalSample, param.causticSample
    done = gt["queries"] - gt["handed_on"][0] - gt["handed_on"][1]
    phot_group = gt["photons_found"] - k_g * gt["handed_on"][0] - k_c * gt["handed_on"][1]
    bytes_total = 96.0 * done + 26.0 * phot_group
    launches = max(gt["launches"], 1.0)
    bytes_launch = bytes_total / launches
    avg_launch_ms = gt["ms_group_kernel"] / (launches / world)   # the slowest rank's sum over its own launches
    achieved = bytes_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    # the two maps' launches of a step run side by side: their bytes over the time from the first begin to the last end
    side = None
    if world == 1 and gt["steps_side_by_side"] > 0 and gt["ms_side_by_side"] > 0:
        side_bw = bytes_total / (gt["ms_side_by_side"] * 1e-3) / 1e9
        side = {"achieved": side_bw, "frac": side_bw / HBM_PEAK_GBS, "ms_per_step": gt["ms_side_by_side"] / gt["steps_side_by_side"],
                "note": "both maps' launches of a step together: their algorithmic bytes over the time from the first "
                        "launch's begin to the last one's end (events on the two streams)"}
    ng, nc, og, oc, phot = (vals["gathers_global"], vals["gathers_caustic"], vals["overflow_global"], vals["overflow_caustic"],
                            vals["gather_photons"])
    ms_group = times["ms_gather_kernel"]
    # the same kernel with the GPU to itself (instrumented iterations after the timed region)
    alone_bytes = 96.0 * ((ng - og) + (nc - oc)) + 26.0 * (phot - k_g * og - k_c * oc)
    alone = alone_bytes / (ms_group * 1e-3) / 1e9 if ms_group > 0 else 0.0
    # HBM traffic comes from PMC passes of their own (FETCH_SIZE, WRITE_SIZE), which cannot run inside this
    # process: the last collection is kept under profiles/ with its provenance.
    traffic, traffic_note, traffic_camera = None, None, None
    tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world == 1 and name == WORKLOAD and not args.size and not args.photons and os.path.exists(tp):
        with open(tp) as f:
            tj = json.load(f)
        traffic, traffic_note = tj.get("bytes_per_launch"), tj.get("note")
        traffic_camera = tj.get("k_camera_paths_bytes_per_launch")
    out = {
        "metric": "SPPM iterations/sec", "value": its, "unit": "iterations/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{name} {W}x{H}, {param.globalSize} global + {param.causticSize} caustic photons/iter, "
                               f"k={k_g}/{k_c}, depth {max_depth}, 1 ray/pixel",
                   "parallelism": (f"photon paths + pixel rows over {world} GPU(s), " +
                                   ("photon maps pulled with peer copies (one process, tkhip_group_*)" if driver == "group"
                                    else "photon all-gather")),
                   "driver": driver, "photon_tree": args.tree},
        "mrays_per_s": vals["rays"] / (ms_step * 1e-3) / 1e6,
        "gathers_per_s": (ng + nc) / (ms_step * 1e-3),
        "photon_paths_per_s": (vals["global_paths"] + vals["caustic_paths"]) / (ms_step * 1e-3),
        "phase_ms": {"photon": times["ms_photon"], "kd_build": times["ms_build"], "camera": times["ms_camera"],
                     "gather": times["ms_gather"], "gather_group_kernel": ms_group},
        "job_ms": job_ms, "iter0_ms": iter0_ms,
        "job": (f"{n_job_iters} iterations (the scene's recurse) from startRaytracing to the resolved image, first iteration "
                f"included, median of {args.jobs} renders; value / ms_per_step are steady-state iterations {args.warmup}.."
                f"{args.warmup + args.steps - 1}" if job_ms is not None else None),
        "roofline": {"bound": "hbm", "kernel": "k_gather_group", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "launches": launches / world, "avg_launch_ms": avg_launch_ms,
                     "algorithmic_bytes_per_launch": bytes_launch,
                     "gathers_per_launch": gt["queries"] / launches, "finished_per_launch": done / launches,
                     "distance_tests_per_gather": gt["photons_staged"] * 64.0 / max(gt["queries"], 1.0),
                     "side_by_side": side,
                     "alone": {"achieved": alone, "frac": alone / HBM_PEAK_GBS, "avg_launch_ms": ms_group,
                               "note": "the same launch in the instrumented iterations after the timed region (the host "
                                       "waits for it before it enqueues the per-query kernels)"},
                     "note": "one launch of k_gather_group per step takes the groups of both maps' gathers"},
    }
    # The whole gather phase (query sort, both gather kernels of both maps, the fused pixel update), from the
    # instrumented iterations: every gather's 56 B query + 26 B per photon found + 40 B pixel slice.
    ms_gather = times["ms_gather"]
    gather_bytes = 96.0 * (ng + nc) + 26.0 * phot
    gp = gather_bytes / (ms_gather * 1e-3) / 1e9 if ms_gather > 0 else 0.0
    out["roofline_gather_phase"] = {"bound": "hbm", "kernels": "k_query_bucket / place / rank, k_sort_queries, k_gather_group, k_gather_wave (both maps); in the first iterations also k_query_bound and k_gather_select",
                                    "achieved": gp, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gp / HBM_PEAK_GBS,
                                    "ms": ms_gather, "algorithmic_bytes": gather_bytes,
                                    "gathers": ng + nc, "photons_found": phot}
    # Path kernels: B_ray = 32 B ray + 48 B hit + 64 B per node of the flattened BVH visited (a node holds both
    # child boxes) + 48 B per primitive tested, the visits and tests counted on the tree that is walked, by a
    # counting instance of the same kernels on the same rays (the instrumented iterations).  The camera lane's
    # rays over the time its two kernels take, the light lanes' over theirs (the lanes run side by side).
    rays_cam, rays_light = vals["rays_camera"], vals["rays"] - vals["rays_camera"]
    c_cam, c_light = vals["counted_rays_camera"], vals["counted_rays"] - vals["counted_rays_camera"]
    if world == 1 and rays_cam > 0 and c_cam > 0 and vals["node_visits_camera"] > 0:
        def lane(rays, counted, nodes, prims, ms, kernels):
            nodes, prims = nodes / max(counted, 1.0) * rays, prims / max(counted, 1.0) * rays   # per ray of the counted iteration
            per_ray = 32.0 + 48.0 + 64.0 * nodes / max(rays, 1.0) + 48.0 * prims / max(rays, 1.0)
            bw = rays * per_ray / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernels": kernels, "rays": rays, "nodes_per_ray": nodes / max(rays, 1.0), "prims_per_ray": prims / max(rays, 1.0),
                    "bytes_per_ray": per_ray, "ms": ms, "achieved": bw, "frac": bw / HBM_PEAK_GBS}
        out["roofline_paths"] = {
            "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "camera": lane(rays_cam, c_cam, vals["node_visits_camera"], vals["prim_tests_camera"], times["ms_camera"],
                           "k_camera_first, k_camera_paths"),
            "light": lane(rays_light, c_light, vals["node_visits_light"], vals["prim_tests_light"], times["ms_photon"],
                          "k_photon_first<0,1>, k_photon_paths<0,1>"),
            "traffic_k_camera_paths": traffic_camera,
            "note": "the three lanes run side by side, so each figure is that lane's rays over its own wall time; node visits "
                    "and primitive tests are counted on the flattened 64-byte-node BVH the kernels walk"}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(scene, W, H, param, max_depth, rr, seed, args.cpu_threads)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)


def gt_dict(ts):
    """Gather totals of the timed region over the ranks (ts: one TkGatherTotals per rank)."""
    return {"launches": float(sum(t.launches for t in ts)), "queries": float(sum(t.queries for t in ts)),
            "handed_on": [float(sum(t.queries_handed_on_k[w] for t in ts)) for w in (0, 1)],
            "photons_found": float(sum(t.photons_found for t in ts)), "photons_staged": float(sum(t.photons_staged for t in ts)),
            "ms_group_kernel": max(t.ms_group_kernel for t in ts),
            "ms_side_by_side": ts[0].ms_group_side_by_side, "steps_side_by_side": ts[0].steps_side_by_side}


def run_group(args):
    """N GPUs from this one process: tkhip_group_* (a context and a host thread per GPU inside the library)."""
    from pathtracer_amd import _abi
    from pathtracer_amd.tracer import SppmGroup
    lib = _abi.load_tkhip()
    present = lib.tkhip_device_count()
    if present < 1:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    world = args.gpus
    if present < world and not args.share_gpus:
        raise SystemExit(f"bench.py --gpus {world} --driver group: {present} GPU(s) present (--share-gpus rehearses on fewer)")
    devices = [r % present for r in range(world)]
    scene, name, W, H, param = load_workload(args)
    max_depth, rr, seed = scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED
    grp = SppmGroup(scene, devices, param, max_depth, rr, W, H, seed=seed)
    ranks_seen = lib.tkhip_group_size(grp.h)
    if ranks_seen != world:
        raise SystemExit(f"bench.py: asked for {world} ranks, the group holds {ranks_seen}")
    ctxs = [lib.tkhip_group_ctx(grp.h, r) for r in range(world)]
    for c in ctxs:
        _abi.check(lib, lib.tkhip_sppm_set_tree(c, 1 if args.tree == "median" else 0), "tkhip_sppm_set_tree")
    for it in range(args.warmup):
        grp.iteration(it)        # returns when every rank has finished the iteration
    for c in ctxs:
        totals_of(lib, c, True)
    last = args.warmup + args.steps - 1
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        grp.iteration(it, begin_next=args.ahead if it < last else 0)
    elapsed = time.perf_counter() - t0
    gts = [totals_of(lib, c, True) for c in ctxs]
    n_inst = 3
    acc = [dict() for _ in range(world)]
    for it in range(args.warmup + args.steps, args.warmup + args.steps + n_inst):
        sts = grp.iteration(it, with_stats=True)
        for r in range(world):
            d = sts[r].as_dict()
            acc[r] = {k: acc[r].get(k, 0) + d[k] for k in d}
    for c in ctxs:
        _abi.check(lib, lib.tkhip_sppm_set_counting(c, 1), "tkhip_sppm_set_counting")
    sts = grp.iteration(args.warmup + args.steps + n_inst, with_stats=True)
    for r in range(world):
        add_counted(acc[r], sts[r].as_dict(), n_inst)
    vals, times = sum_stats(acc, n_inst)
    n_job_iters = num_recurse_of(lib, ctxs[0])
    grp.close()
    report(args, name, scene, W, H, param, max_depth, rr, seed, world, ranks_seen, "group", elapsed, vals, times, gt_dict(gts),
           None, None, n_job_iters)


def num_recurse_of(lib, ctx):
    from pathtracer_amd import _abi
    nr = C.c_int32()
    _abi.check(lib, lib.tkhip_sppm_counters(ctx, None, None, C.byref(nr)), "tkhip_sppm_counters")
    return nr.value


def run_ranks(args):
    """This process is one rank (of WORLD_SIZE, or the only one)."""
    import torch
    import torch.distributed as dist

    from pathtracer_amd import _abi
    from pathtracer_amd import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if args.backend != "nccl" or args.share_gpus:
        local_rank %= max(torch.cuda.device_count(), 1)   # rehearsal: ranks may share a GPU
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if world > 1:
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else "cpu"
    # every rank that was asked for has come up and takes part in the collectives
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())
        if ranks_seen != world:
            raise SystemExit(f"bench.py: {ranks_seen} of {world} ranks took part in the all-reduce")

    scene, name, W, H, param = load_workload(args)
    max_depth, rr, seed = scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED
    backend = D.HipBackend(scene, param, max_depth, rr, seed, W, H, rank, world, local_rank)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    lib, ctx_h = backend.lib, backend.h
    _abi.check(lib, lib.tkhip_sppm_set_tree(ctx_h, 1 if args.tree == "median" else 0), "tkhip_sppm_set_tree")

    for it in range(args.warmup):
        D.iteration(backend, it, world)
    sync()
    totals_of(lib, ctx_h, True)   # the grouped gather kernel is event-timed on its stream at every launch: count from here
    t0 = time.perf_counter()
    last = args.warmup + args.steps - 1
    for it in range(args.warmup, args.warmup + args.steps):
        # several ranks, --ahead 1 / 2: the light paths (and the camera paths) of the next step are begun under
        # this step's gathers (not after the last one: nothing of step K+1 is left in flight)
        D.iteration(backend, it, world, begin_next=args.ahead if it < last else 0)
    sync()
    elapsed = time.perf_counter() - t0
    gt = totals_of(lib, ctx_h, True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # instrumented iterations after the timed region: counters, per-phase and per-kernel device time
    acc = {}
    n_inst = 3
    for it in range(args.warmup + args.steps, args.warmup + args.steps + n_inst):
        st = _abi.TkIterStats()
        D.iteration(backend, it, world, st)
        d = st.as_dict()
        acc = {k: acc.get(k, 0) + d[k] for k in d}
    sync()
    # one more iteration with the counting instances of the path kernels: node visits and primitive tests per ray
    _abi.check(lib, lib.tkhip_sppm_set_counting(ctx_h, 1), "tkhip_sppm_set_counting")
    st = _abi.TkIterStats()
    D.iteration(backend, args.warmup + args.steps + n_inst, world, st)
    add_counted(acc, st.as_dict(), n_inst)
    _abi.check(lib, lib.tkhip_sppm_set_counting(ctx_h, 0), "tkhip_sppm_set_counting")
    sync()
    gsum = [gt.launches, gt.queries, gt.queries_handed_on_k[0], gt.queries_handed_on_k[1], gt.photons_found, gt.photons_staged]
    vsum = torch.tensor([acc[k] for k in STAT_KEYS] + gsum, dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([acc[k] for k in TIME_KEYS] + [gt.ms_group_kernel], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    vsum, tmax = vsum.tolist(), tmax.tolist()
    vals = {k: vsum[i] / n_inst for i, k in enumerate(STAT_KEYS)}
    times = {k: tmax[i] / n_inst for i, k in enumerate(TIME_KEYS)}
    g = vsum[len(STAT_KEYS):]
    gtd = {"launches": g[0], "queries": g[1], "handed_on": [g[2], g[3]], "photons_found": g[4], "photons_staged": g[5],
           "ms_group_kernel": tmax[len(TIME_KEYS)], "ms_side_by_side": gt.ms_group_side_by_side,
           "steps_side_by_side": gt.steps_side_by_side}

    # The job the reference times (src/main.cpp:51-58): a whole render of the scene's own iteration count,
    # first iteration included, from startRaytracing to the resolved image.  Repeated so that the GPU leg of
    # this program lasts seconds, not a fraction of one.
    job_ms = iter0_ms = None
    n_job_iters = backend.num_recurse
    if world == 1 and args.jobs > 0:
        backend.close()
        jobs, firsts = [], []
        for _ in range(args.jobs):
            backend = D.HipBackend(scene, param, max_depth, rr, seed, W, H, rank, world, local_rank)
            _abi.check(backend.lib, backend.lib.tkhip_sppm_set_tree(backend.h, 1 if args.tree == "median" else 0), "tkhip_sppm_set_tree")
            t_b = time.perf_counter()
            backend.iteration_single(0)
            t_c = time.perf_counter()
            for it in range(1, n_job_iters):
                backend.iteration_single(it)
            backend.resolve()
            t_d = time.perf_counter()
            backend.close()
            jobs.append((t_d - t_b) * 1e3)
            firsts.append((t_c - t_b) * 1e3)
        job_ms = float(np.median(jobs))
        iter0_ms = float(np.median(firsts))
        backend = None

    if rank == 0:
        report(args, name, scene, W, H, param, max_depth, rr, seed, world, ranks_seen, "torch", elapsed, vals, times, gtd,
               job_ms, iter0_ms, n_job_iters)
    if backend is not None:
        backend.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    cmd = launcher_command(args.gpus, args.driver, os.environ, argv)
    if cmd is not None:
        # Nothing in this process has touched a GPU (torch is not even imported): the ranks are children of
        # the launcher, and this process only passes their output and exit status on.
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    if args.gpus > 1 and args.driver == "group" and "WORLD_SIZE" not in os.environ:
        run_group(args)
    else:
        run_ranks(args)


if __name__ == "__main__":
    main()
