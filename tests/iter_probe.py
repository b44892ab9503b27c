"""Per-iteration statistics of a render (profiling helper, not a test): gathers, queries handed to the per-query
kernel, distance tests and photons found per gather, phase times.  Usage:
    python tests/iter_probe.py [scene] [iterations] [--synthetic QUADS --size N --photons G C]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pathtracer_amd import _abi  # noqa: E402
from pathtracer_amd import distributed as D  # noqa: E402
from pathtracer_amd.scene import Scene, asset_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="cornell_dragon.scene")
    ap.add_argument("iterations", nargs="?", type=int, default=16)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--photons", type=int, nargs=2, default=None)
    ap.add_argument("--plain", action="store_true", help="no statistics: iterations as a render runs them, wall time only")
    args = ap.parse_args()
    import time
    import torch
    scene = Scene.synthetic(args.synthetic) if args.synthetic else Scene.load(asset_scene(args.scene))
    W = args.size or scene.width
    H = args.size or scene.height
    param = scene.sppm_param(globalSize=args.photons[0], causticSize=args.photons[1]) if args.photons else scene.sppm_param()
    backend = D.HipBackend(scene, param, scene.tracer.max_depth, scene.tracer.russian_roulette, 0x5EED, W, H, 0, 1, 0)
    for it in range(args.iterations):
        if args.plain:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            backend.iteration_single(it)
            torch.cuda.synchronize()
            print("iter %2d: %.2f ms" % (it, (time.perf_counter() - t0) * 1e3), flush=True)
            continue
        st = _abi.TkIterStats()
        backend.iteration_single(it, st)
        d = st.as_dict()
        ng, nc = d["gathers_global"], d["gathers_caustic"]
        nq = max(ng + nc, 1)
        print("iter %2d: total %.2f ms photon %.2f build %.2f camera %.2f gather %.2f (group kernels %.2f) | gathers %d + %d, "
              "handed on %d + %d, tests/gather %.0f, found/gather %.1f" % (
                  it, d["ms_total"], d["ms_photon"], d["ms_build"], d["ms_camera"], d["ms_gather"], d["ms_gather_kernel"],
                  ng, nc, d["overflow_global"], d["overflow_caustic"], d["group_staged"] * 64.0 / nq,
                  d["gather_photons"] / nq), flush=True)
    backend.close()


if __name__ == "__main__":
    main()
