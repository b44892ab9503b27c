"""Times an iteration driven phase by phase (the multi-rank code path) on one rank (profiling helper, not a test)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import distributed as D
from pathtracer_amd.scene import Scene, asset_scene

scene = Scene.load(asset_scene("cornell_dragon.scene"))
param = scene.sppm_param()
for world in (1, 2, 8):
    be = D.HipBackend(scene, param, 20, 0.8, 0x5EED, 1024, 1024, 0, world, 0)
    def step(it):
        be.camera_begin(it)
        be.photon_pass(it)
        if world > 1:   # this rank's share only: install it as if it were the whole map
            for which in (0, 1):
                pos, power, d, n, shot = be.export_local(which)
                be.set_map(which, pos, power, d, n, shot)
        be.build_maps()
        be.camera_pass(it)
    for it in range(5):
        step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(5, 25):
        step(it)
    torch.cuda.synchronize()
    # one more iteration, phase by phase with a synchronisation after each
    ph = {}
    def timed(name, fn):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
        ph[name] = ph.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    for it in range(25, 28):
        timed("photon_pass", lambda: be.photon_pass(it))
        timed("build", lambda: be.build_maps())
        timed("camera_begin+wait", lambda: (be.camera_begin(it), torch.cuda.synchronize()))
        timed("camera_pass(gathers)", lambda: be.camera_pass(it))
    print("   phases alone, ms:", {k: round(v / 3, 2) for k, v in ph.items()})
    print("rank 0 of", world, ":", (time.perf_counter() - t0) / 20 * 1e3, "ms per iteration, phase by phase (no collective)")
    be.close()
