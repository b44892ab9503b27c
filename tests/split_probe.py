"""Times one rank of a multi-rank render on one GPU (profiling helper, not a test): the rank runs
distributed.iteration() as rank 0 of `world`, with the all-gather replaced by `world` copies of its own buffer, so
that it traces its share of the paths and builds and gathers maps of the full size.

    python3 tests/split_probe.py [--photons G C] [--synthetic QUADS --size N] [--iters A B] world [world ...]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import distributed as D
from pathtracer_amd.scene import Scene, asset_scene

ap = argparse.ArgumentParser()
ap.add_argument("worlds", type=int, nargs="*", default=[1, 2, 4, 8])
ap.add_argument("--photons", type=int, nargs=2, default=None)
ap.add_argument("--synthetic", type=int, default=0)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, nargs=2, default=[5, 20], metavar=("WARMUP", "TIMED"))
args = ap.parse_args()

D._all_gather_flat = lambda t, world: t.repeat(world)
scene = Scene.synthetic(args.synthetic) if args.synthetic else Scene.load(asset_scene("cornell_dragon.scene"))
param = scene.sppm_param(globalSize=args.photons[0], causticSize=args.photons[1]) if args.photons else scene.sppm_param()
warm, timed_n = args.iters
for world in args.worlds:
    assert param.globalSize % world == 0 and param.causticSize % world == 0, "copies of rank 0's quota must add up to the map sizes"
    be = D.HipBackend(scene, param, 20, 0.8, 0x5EED, args.size, args.size, 0, world, 0)
    if world == 1:
        be_step = lambda it: be.iteration_single(it)
    else:
        ahead = int(os.environ.get("SPLIT_PROBE_AHEAD", "0"))   # 0: nothing ahead, 1: the light paths, 2: light and camera paths
        be_step = lambda it: D.iteration(be, it, world, begin_next=ahead if warm <= it < warm + timed_n - 1 else 0)
    for it in range(warm):
        be_step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(warm, warm + timed_n):
        be_step(it)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / timed_n * 1e3
    ph = {}
    def timed(name, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    if world > 1:
        for it in range(warm + timed_n, warm + timed_n + 3):
            timed("photon pass", lambda: be.photon_pass(it))
            timed("exchange", lambda: D.exchange_photons(be, world))
            timed("builds", lambda: be.build_maps())
            timed("camera paths", lambda: (be.camera_begin(it), torch.cuda.synchronize()))
            timed("gathers", lambda: be.camera_pass(it))
    print("rank 0 of %d: %.2f ms per iteration; phases alone, ms: %s" % (world, ms, {k: round(v / 3, 2) for k, v in ph.items()}), flush=True)
    be.close()
