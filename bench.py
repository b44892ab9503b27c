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
    return {
        "value": 1.0 / ref["sec_iter"], "unit": "iterations/s", "cores": threads, "kind": "port", "iteration": 0,
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


def main():
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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo rehearses N > 1 on fewer GPUs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pathtracer_amd import _abi
    from pathtracer_amd import distributed as D
    from pathtracer_amd.scene import Scene, asset_scene

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)   # rehearsal: ranks may share a GPU
    if world > 1:
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = torch.device("cuda", local_rank)

    scene = Scene.synthetic(args.synthetic) if args.synthetic else Scene.load(asset_scene(args.scene))
    if args.synthetic:
        args.scene = f"synthetic height field {args.synthetic}x{args.synthetic} quads ({scene.n_triangles} triangles)"
    W = args.size or scene.width
    H = args.size or scene.height
    param = scene.sppm_param(globalSize=args.photons[0], causticSize=args.photons[1]) if args.photons else scene.sppm_param()
    max_depth, rr, seed = scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED
    backend = D.HipBackend(scene, param, max_depth, rr, seed, W, H, rank, world, local_rank)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    lib, ctx_h = backend.lib, backend.h
    _abi.check(lib, lib.tkhip_sppm_set_tree(ctx_h, 1 if args.tree == "median" else 0), "tkhip_sppm_set_tree")

    def gather_totals(reset):
        t = _abi.TkGatherTotals()
        _abi.check(lib, lib.tkhip_sppm_gather_totals(ctx_h, C.byref(t), 1 if reset else 0), "tkhip_sppm_gather_totals")
        return t

    for it in range(args.warmup):
        D.iteration(backend, it, world)
    sync()
    gather_totals(True)   # the grouped gather kernel is event-timed on its stream at every launch: count from here
    t0 = time.perf_counter()
    last = args.warmup + args.steps - 1
    for it in range(args.warmup, args.warmup + args.steps):
        # several ranks, --ahead 1 / 2: the light paths (and the camera paths) of the next step are begun under
        # this step's gathers (not after the last one: nothing of step K+1 is left in flight)
        D.iteration(backend, it, world, begin_next=args.ahead if world > 1 and it < last else 0)
    sync()
    elapsed = time.perf_counter() - t0
    gt = gather_totals(True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # instrumented iterations after the timed region: counters, per-phase and per-kernel device time
    acc = None
    n_inst = 3
    for it in range(args.warmup + args.steps, args.warmup + args.steps + n_inst):
        st = _abi.TkIterStats()
        D.iteration(backend, it, world, st)
        d = st.as_dict()
        acc = d if acc is None else {k: acc[k] + d[k] for k in d}
    sync()
    vals = torch.tensor([acc["rays"], acc["gathers_global"], acc["gathers_caustic"], acc["gather_photons"],
                         acc["overflow_global"], acc["overflow_caustic"], acc["global_paths"], acc["caustic_paths"],
                         acc["group_staged"],
                         n_inst * gt.queries, n_inst * gt.queries_handed_on_k[0], n_inst * gt.queries_handed_on_k[1],
                         n_inst * gt.photons_found, n_inst * gt.photons_staged, n_inst * gt.launches, acc["rays_camera"]],
                        dtype=torch.float64, device=dev)
    times = torch.tensor([acc["ms_gather_kernel"], acc["ms_photon"], acc["ms_build"], acc["ms_camera"], acc["ms_gather"],
                          n_inst * gt.ms_group_kernel],
                         dtype=torch.float64, device=dev)
    if world > 1:
        if args.backend != "nccl":
            vals, times = vals.cpu(), times.cpu()
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    vals = (vals / n_inst).tolist()
    times = (times / n_inst).tolist()
    rays, ng, nc, phot, og, oc, gpaths, cpaths, staged, t_q, t_og, t_oc, t_phot, t_staged, t_launches, rays_cam = vals
    ms_group, ms_photon, ms_build, ms_camera, ms_gather, t_ms = times

    # The job the reference times (src/main.cpp:51-58): a whole render of the scene's own iteration count,
    # first iteration included, from startRaytracing to the resolved image.  Repeated so that the GPU leg of
    # this program lasts seconds, not a fraction of one.
    job_ms = iter0_ms = None
    n_job_iters = backend.num_recurse
    if world == 1 and args.jobs > 0:
        backend.close()
        jobs, firsts = [], []
        for _ in range(args.jobs):
            t_a = time.perf_counter()
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
        its = args.steps / elapsed
        ms_step = elapsed / args.steps * 1e3
        # Algorithmic bytes of a gather: 56 B query + 26 B per photon found + 40 B pixel slice.
        # The grouped kernel finishes the gathers that met fewer than k photons; the others end
        # with exactly k photons in the per-query kernel and are not charged to it here.
        # Measured over the timed region: every launch of the kernel is bracketed by HIP events on
        # the stream it runs on (the two maps' launches of a step run side by side there).
        k_g, k_c = param.globalSample, param.causticSample
        done = t_q - t_og - t_oc
        phot_group = t_phot - k_g * t_og - k_c * t_oc
        bytes_total = 96.0 * done + 26.0 * phot_group
        launches = max(t_launches, 1.0)
        bytes_launch = bytes_total / launches
        avg_launch_ms = t_ms / (launches / world)   # t_ms is the slowest rank's sum over its own launches
        achieved = bytes_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        # the two maps' launches of a step run side by side: their bytes over the time from the first begin to the last end
        side = None
        if world == 1 and gt.steps_side_by_side > 0 and gt.ms_group_side_by_side > 0:
            side_ms = gt.ms_group_side_by_side / gt.steps_side_by_side
            side_bw = bytes_total / (gt.ms_group_side_by_side * 1e-3) / 1e9
            side = {"achieved": side_bw, "frac": side_bw / HBM_PEAK_GBS, "ms_per_step": side_ms,
                    "note": "both maps' launches of a step together: their algorithmic bytes over the time from the first "
                            "launch's begin to the last one's end (events on the two streams)"}
        # the same kernel with the GPU to itself (instrumented iterations after the timed region)
        alone_bytes = 96.0 * ((ng - og) + (nc - oc)) + 26.0 * (phot - k_g * og - k_c * oc)
        alone = alone_bytes / (ms_group * 1e-3) / 1e9 if ms_group > 0 else 0.0
        # HBM traffic of the kernel comes from PMC passes of their own (FETCH_SIZE, WRITE_SIZE), which cannot
        # run inside this process: the last collection is kept under profiles/ with its provenance.
        traffic, traffic_note = None, None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if world == 1 and args.scene == WORKLOAD and not args.size and os.path.exists(tp):
            with open(tp) as f:
                tj = json.load(f)
            traffic, traffic_note = tj.get("bytes_per_launch"), tj.get("note")
        out = {
            "metric": "SPPM iterations/sec", "value": its, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {param.globalSize} global + {param.causticSize} caustic photons/iter, "
                                   f"k={k_g}/{k_c}, depth {max_depth}, 1 ray/pixel",
                       "parallelism": f"photon paths + pixel rows over {world} GPU(s), photon all-gather",
                       "photon_tree": args.tree},
            "mrays_per_s": rays / (ms_step * 1e-3) / 1e6,
            "gathers_per_s": (ng + nc) / (ms_step * 1e-3),
            "photon_paths_per_s": (gpaths + cpaths) / (ms_step * 1e-3),
            "phase_ms": {"photon": ms_photon, "kd_build": ms_build, "camera": ms_camera, "gather": ms_gather,
                         "gather_group_kernel": ms_group},
            "job_ms": job_ms, "iter0_ms": iter0_ms,
            "job": (f"{n_job_iters} iterations (the scene's recurse) from startRaytracing to the resolved image, first iteration "
                    f"included, median of {args.jobs} renders; value / ms_per_step are steady-state iterations {args.warmup}.."
                    f"{args.warmup + args.steps - 1}" if job_ms is not None else None),
            "roofline": {"bound": "hbm", "kernel": "k_gather_group", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "launches": launches / world, "avg_launch_ms": avg_launch_ms,
                         "algorithmic_bytes_per_launch": bytes_launch,
                         "gathers_per_launch": t_q / launches, "finished_per_launch": done / launches,
                         "distance_tests_per_gather": t_staged * 64.0 / max(t_q, 1.0),
                         "side_by_side": side,
                         "alone": {"achieved": alone, "frac": alone / HBM_PEAK_GBS, "avg_launch_ms": ms_group / 2.0,
                                   "note": "the same kernel with the GPU to itself (instrumented iterations after the "
                                           "timed region, the two maps one after the other)"}},
        }
        # The whole gather phase (query sort, both gather kernels of both maps, the fused pixel update), from the
        # instrumented iterations: every gather's 56 B query + 26 B per photon found + 40 B pixel slice.
        gather_bytes = 96.0 * (ng + nc) + 26.0 * phot
        gp = gather_bytes / (ms_gather * 1e-3) / 1e9 if ms_gather > 0 else 0.0
        out["roofline_gather_phase"] = {"bound": "hbm", "kernels": "k_query_bucket / place / rank, k_sort_queries, k_gather_group, k_gather_wave (both maps); in the first iterations also k_query_bound and k_gather_select",
                                        "achieved": gp, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gp / HBM_PEAK_GBS,
                                        "ms": ms_gather, "algorithmic_bytes": gather_bytes,
                                        "gathers": ng + nc, "photons_found": phot}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, param, max_depth, rr, seed, args.cpu_threads)
            # Path kernels: B_ray = 32 B ray + 48 B hit + 32 B per box test + 48 B per primitive test, the tests
            # per ray counted by the CPU port on the same rays (implementation independent); the camera lane's
            # rays over the time its two kernels take, the light lanes' over theirs (the lanes run side by side).
            pr = out["cpu_baseline"]["per_ray"]
            def b_ray(c):
                return 32.0 + 48.0 + 32.0 * c["box_tests_per_ray"] + 48.0 * c["prim_tests_per_ray"]
            cam = rays_cam * b_ray(pr["tc_camera"]) / (ms_camera * 1e-3) / 1e9 if ms_camera > 0 else 0.0
            lig = (rays - rays_cam) * b_ray(pr["tc_photon"]) / (ms_photon * 1e-3) / 1e9 if ms_photon > 0 else 0.0
            out["roofline_paths"] = {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "camera": {"kernels": "k_camera_first, k_camera_paths", "rays": rays_cam, "bytes_per_ray": b_ray(pr["tc_camera"]),
                           "ms": ms_camera, "achieved": cam, "frac": cam / HBM_PEAK_GBS},
                "light": {"kernels": "k_photon_first<0,1>, k_photon_paths<0,1>", "rays": rays - rays_cam,
                          "bytes_per_ray": b_ray(pr["tc_photon"]), "ms": ms_photon, "achieved": lig, "frac": lig / HBM_PEAK_GBS},
                "note": "the three lanes run side by side, so each figure is that lane's rays over its own wall time"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if backend is not None:
        backend.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
