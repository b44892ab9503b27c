"""Times one rank of a multi-rank render on one GPU (profiling helper, not a test): the rank runs
distributed.iteration() as rank 0 of `world`, with the all-gather replaced by `world` copies of its own table, so
that it traces its share of the paths and builds and gathers maps of the full size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import distributed as D
from pathtracer_amd.scene import Scene, asset_scene

D._all_gather = lambda t, world: [t] * world
scene = Scene.load(asset_scene("cornell_dragon.scene"))
param = scene.sppm_param()
for world in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]):
    be = D.HipBackend(scene, param, 20, 0.8, 0x5EED, 1024, 1024, 0, world, 0)
    if world == 1:
        be_step = lambda it: be.iteration_single(it)
    else:
        ahead = int(os.environ.get("SPLIT_PROBE_AHEAD", "0"))   # 0: nothing ahead, 1: the light paths, 2: light and camera paths
        be_step = lambda it: D.iteration(be, it, world, begin_next=ahead if 5 <= it < 24 else 0)
    for it in range(5):
        be_step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(5, 25):
        be_step(it)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ph = {}
    def timed(name, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize()
        ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t) * 1e3
        return r
    if world > 1:
        for it in range(25, 28):
            timed("photon pass", lambda: be.photon_pass(it))
            timed("exchange", lambda: D.exchange_photons(be, world))
            timed("builds", lambda: be.build_maps())
            timed("camera paths", lambda: (be.camera_begin(it), torch.cuda.synchronize()))
            timed("gathers", lambda: be.camera_pass(it))
    print("rank 0 of %d: %.2f ms per iteration; phases alone, ms: %s" % (world, ms, {k: round(v / 3, 2) for k, v in ph.items()}), flush=True)
    be.close()
