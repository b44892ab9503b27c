"""Throughput of the two other tracers on the same path kernels (a helper, not a test): the hash-grid SPPM variant
(tk::SPPM) and the path tracer (tk::PathTracer), cornell_dragon at 1024^2 unless another scene / size is named.
    python tests/variants_probe.py [scene] [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd.scene import Scene, asset_scene  # noqa: E402
from pathtracer_amd.tracer import HashGridSppm, PathTracer  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_dragon.scene"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    scene = Scene.load(asset_scene(name))
    prm = scene.sppm_param(causticSize=0)
    hg = HashGridSppm(scene, prm, 1, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W)
    for it in range(4):
        hg.iteration(it)
    t, rays = time.perf_counter(), 0
    for it in range(4, 20):
        rays += hg.iteration(it).rays
    ms = (time.perf_counter() - t) / 16 * 1e3
    print(f"hash-grid SPPM, {name} {W}x{W}, {prm.globalSize} photon paths per iteration: {ms:.2f} ms per iteration, "
          f"{rays / 16 / ms / 1e3:.0f} Mrays/s", flush=True)
    hg.close()
    spp = 16
    pt = PathTracer(scene, 2 * spp, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W)   # (a power of two)
    pt.render_samples(spp)
    t = time.perf_counter()
    st = pt.render_samples(spp)
    ms = (time.perf_counter() - t) * 1e3
    print(f"path tracer, {name} {W}x{W}: {ms / spp:.2f} ms per sample of every pixel, {W * W * spp / ms / 1e3:.0f} M samples/s, "
          f"{st.rays / ms / 1e3:.0f} Mrays/s", flush=True)
    pt.close()


if __name__ == "__main__":
    main()
