"""Full-size parity over many iterations (a helper, not a test: the oracle needs seconds per iteration):
    python tests/long_parity.py cornell_dragon.scene 1024 12
runs the scene at its own photon counts on the GPU and in the CPU oracle side by side and compares counters every
iteration, the pixel state every third one and the resolved images at the end."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pathtracer_amd.scene import Scene, asset_scene
from tests.test_parity_gpu import _tracers, _assert_pixels_equal
name, size, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
scene = Scene.load(asset_scene(name))
prm = scene.tracer.sppm
tr, osc, orr = _tracers(scene, size, prm.globalSize, prm.causticSize, 16)
for it in range(iters):
    t = time.time()
    st, ost = tr.iteration(), orr.iteration(it)
    for f in ("global_paths", "caustic_paths", "global_stored", "caustic_stored", "gathers_global", "gathers_caustic"):
        assert getattr(st, f) == getattr(ost, f), (it, f, getattr(st, f), getattr(ost, f))
    if it % 3 == 2 or it == iters - 1:
        g, o = tr.pixels(), orr.pixels()
        _assert_pixels_equal(g, o)
        print("iteration", it, "equal; NaN pixels", int(np.isnan(g).any(axis=1).sum()), "overflow", st.overflow_global, st.overflow_caustic, round(time.time() - t, 1), "s", flush=True)
gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
print("image max diff", np.abs(gi - oi).max(), "equal share", (gi == oi).mean())
