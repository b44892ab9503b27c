"""Calibration of bench.py's cpu_baseline (the oracle in its reference-faithful mode) against the reference
itself: seconds per iteration of upstream's SppmTracer::render (oracle/_ref/ref_probe2, mode "render": its own
threads, its own random numbers) and of the oracle on the same scene, image size, photon counts and thread count.
Runs where /root/reference exists and `bash oracle/build_ref.sh` has been run; a helper, not a test.

    python3 oracle/calibrate_cpu_baseline.py [--scene cornell_dragon.scene] [--size 256] [--threads 8]

The reference's time per iteration is the difference between a render of 3 and a render of 1 iterations (scene
loading, BVH build and image write cancel out), halved."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from make_ref_golden import PROBE2, SCENES_DIR, write_camera   # noqa: E402


def reference_seconds(scene, file, size, gsize, csize, recurse, threads):
    with tempfile.TemporaryDirectory() as tmp:
        ind, outd = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(ind)
        os.makedirs(outd)
        write_camera(scene, ind)
        t0 = time.time()
        subprocess.run([PROBE2, "render", ind, outd, file, "cal", str(size), str(size), str(recurse), str(gsize), str(csize), "1",
                        str(threads)], cwd=SCENES_DIR, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return time.time() - t0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_dragon.scene")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--photons", type=int, nargs=2, default=(20000, 5000))
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args(argv)
    if not os.path.exists(PROBE2):
        raise SystemExit(f"{PROBE2} missing: run `bash oracle/build_ref.sh` first")
    from pathtracer_amd.scene import Scene, asset_scene
    from oracle import oracle_py as O
    import bench
    scene = Scene.load(asset_scene(args.scene))
    g, c = args.photons
    t1 = reference_seconds(scene, args.scene, args.size, g, c, 1, args.threads)
    t3 = reference_seconds(scene, args.scene, args.size, g, c, 3, args.threads)
    ref_iter = (t3 - t1) / 2.0
    param = scene.sppm_param(globalSize=g, causticSize=c)
    osc = O.OracleScene(scene, args.size, args.size)
    faithful = bench._oracle_iteration(O, osc, param, scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED, args.threads, args.size, 1e9,
                                       split_mode=0, faithful_bvh=1, rng_cost=1)
    fair = bench._oracle_iteration(O, osc, param, scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED, args.threads, args.size, 1e9,
                                   split_mode=1, faithful_bvh=0, rng_cost=0)
    osc.close()
    print(json.dumps({"scene": args.scene, "size": args.size, "photons": [g, c], "threads": args.threads,
                      "reference_render_1_iteration_s": t1, "reference_render_3_iterations_s": t3,
                      "reference_s_per_iteration": ref_iter,
                      "oracle_faithful_s_per_iteration": faithful["sec_iter"], "oracle_fair_s_per_iteration": fair["sec_iter"],
                      "faithful_over_reference": faithful["sec_iter"] / ref_iter}, indent=1))


if __name__ == "__main__":
    main()
