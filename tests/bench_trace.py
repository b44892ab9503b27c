"""Ray-traversal microbenchmark (profiling helper, not a test): traces three ray sets on the dragon scene through
tkhip_trace_rays; run it under `rocprofv3 --kernel-trace --stats` and read the k_trace_rays durations
(launch order: primary, diffuse bounce, towards the light)."""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import Context


def main():
    sc = Scene.load(asset_scene(sys.argv[1] if len(sys.argv) > 1 else "cornell_dragon.scene"))
    ctx = Context(0)
    ctx.set_scene(sc)
    W = H = 1024
    ctx.set_camera(sc, W, H)
    ys, xs = np.mgrid[0:H, 0:W]
    raster = np.stack([xs.ravel() + 0.5, ys.ravel() + 0.5], 1).astype(np.float32)
    org, d = ctx.camera_rays(raster)
    for rep in range(2):
        prim, t, p, n, perr = ctx.trace_rays(org, d)
    hit = prim >= 0
    rng = np.random.default_rng(1)
    m = int(hit.sum())
    # cosine-ish random directions about the normal facing the viewer
    nn = n[hit] * np.sign(-(n[hit] * d[hit]).sum(1, keepdims=True))
    r = rng.normal(size=(m, 3)).astype(np.float32)
    r /= np.linalg.norm(r, axis=1, keepdims=True)
    d2 = nn + 0.999 * r
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
    o2 = (p[hit] + 1e-3 * nn).astype(np.float32)
    for rep in range(2):
        prim2 = ctx.trace_rays(o2, d2)[0]
    light = np.array([0.0, 2.18, 2.2], np.float32) if False else None
    # towards a point near the ceiling light of the Cornell scenes
    tri = sc.triangles_np()
    lt = sc.lights[0]
    ob = sc.objects[lt.object]
    c = tri[ob.first:ob.first + ob.count].reshape(-1, 3).mean(0).astype(np.float32)
    d3 = c[None, :] - o2
    d3 = (d3 / np.linalg.norm(d3, axis=1, keepdims=True)).astype(np.float32)
    for rep in range(2):
        prim3 = ctx.trace_rays(o2, d3)[0]
    # inside the dragon: from its surface inward
    dr = prim[hit] >= 12
    o4 = (p[hit][dr] - 1e-3 * nn[dr]).astype(np.float32)
    d4 = -nn[dr] + 0.7 * r[dr]
    d4 = (d4 / np.linalg.norm(d4, axis=1, keepdims=True)).astype(np.float32)
    reps = int(np.ceil(len(o2) / max(len(o4), 1)))
    o4 = np.tile(o4, (reps, 1))[:len(o2)].copy()
    d4 = np.tile(d4, (reps, 1))[:len(o2)].copy()
    for rep in range(2):
        prim4, t4 = ctx.trace_rays(o4, d4)[:2]
    print("inside: hits", int((prim4 >= 0).sum()), "mean t", float(t4[prim4 >= 0].mean()))
    print("rays", len(org), m, m, "hits", int(hit.sum()), int((prim2 >= 0).sum()), int((prim3 >= 0).sum()))


if __name__ == "__main__":
    main()
