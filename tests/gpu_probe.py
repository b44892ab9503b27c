"""Exploratory GPU run (not collected by pytest): exercises every kernel-level
entry point against the CPU oracle and prints timings.  Usage:
    python tests/gpu_probe.py [scene] [width]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import oracle_py as O  # noqa: E402
from pathtracer_amd.scene import Scene, asset_scene  # noqa: E402
from pathtracer_amd.tracer import Context, PhotonMap, SppmTracer  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cornell_sphere.scene"
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    rng = np.random.default_rng(1)
    scene = Scene.load(asset_scene(name))
    print("scene", name, "tris", scene.n_triangles, "spheres", scene.n_spheres, flush=True)
    ctx = Context(0)
    t0 = time.time()
    ctx.set_scene(scene)
    print("set_scene %.3fs" % (time.time() - t0), flush=True)
    ctx.set_camera(scene, W, W)
    osc = O.OracleScene(scene, W, W)

    # radical inverse
    dim = rng.integers(0, 7, 4096).astype(np.int32)
    idx = rng.integers(0, 1 << 40, 4096).astype(np.uint64)
    idx[:64] = np.arange(64)
    g = ctx.radical_inverse(dim, idx)
    o = np.array([O.lib().orc_radical_inverse(int(d), int(i)) for d, i in zip(dim, idx)], np.float32)
    print("radical_inverse mismatches:", int((g != o).sum()), flush=True)

    # camera rays
    raster = rng.random((4096, 2)).astype(np.float32) * W
    go, gd = ctx.camera_rays(raster)
    oo, od = osc.camera_rays(raster)
    print("camera rays max diff", np.abs(go - oo).max(), np.abs(gd - od).max(), flush=True)

    # rays
    n = 200000
    org = np.tile(go[:1], (n, 1))
    ras = rng.random((n, 2)).astype(np.float32) * W
    _, dirs = osc.camera_rays(ras)
    t0 = time.time()
    gp, gt, gpp, gn, ge = ctx.trace_rays(org, dirs)
    t1 = time.time()
    op, ot, opp, on, oe = osc.trace(org, dirs, faithful=0)
    t2 = time.time()
    hit = op >= 0
    print("trace: gpu %.3fs oracle %.3fs hits %d prim mismatches %d" % (t1 - t0, t2 - t1, hit.sum(), (gp != op).sum()), flush=True)
    same = (gp == op) & hit
    if same.any():
        print("  t maxrel", (np.abs(gt[same] - ot[same]) / ot[same]).max(), "p maxabs", np.abs(gpp[same] - opp[same]).max(),
              "n maxabs", np.abs(gn[same] - on[same]).max(), "perr maxabs", np.abs(ge[same] - oe[same]).max(), flush=True)
    # secondary rays from hit points (random directions)
    m = int(hit.sum())
    d2 = rng.normal(size=(m, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (opp[hit] + 1e-3 * on[hit] * np.sign((d2 * on[hit]).sum(1, keepdims=True))).astype(np.float32)
    gp2, gt2, *_ = ctx.trace_rays(o2, d2)
    op2, ot2, *_ = osc.trace(o2, d2, faithful=0)
    print("secondary: prim mismatches %d of %d" % ((gp2 != op2).sum(), m), flush=True)
    opf, *_ = osc.trace(o2[:20000], d2[:20000], faithful=1)
    print("oracle faithful vs fast traversal mismatches:", int((opf != op2[:20000]).sum()), flush=True)

    # photon map gather
    N, k = 20000, 100
    pos = (rng.random((N, 3)).astype(np.float32) * 4 - 2)
    pos[:, 1] = np.where(rng.random(N) < 0.5, -2.0, pos[:, 1]).astype(np.float32)
    power = rng.random((N, 3)).astype(np.float32)
    th = rng.integers(0, 256, N).astype(np.uint8)
    ph = rng.integers(0, 256, N).astype(np.uint8)
    gm = PhotonMap(ctx, N, k)
    gm.store(pos, power, th, ph)
    t0 = time.time()
    gm.build()
    print("kd build %.3fs" % (time.time() - t0), gm.tree_info()[:3], flush=True)
    om = O.OracleMap(N, k)
    om.store(pos, power, th, ph)
    om.build(0)
    nq = 5000
    qp = (rng.random((nq, 3)).astype(np.float32) * 4 - 2)
    qp[: nq // 2, 1] = -2.0
    qn = np.tile(np.array([[0, 1, 0]], np.float32), (nq, 1))
    qwo = rng.normal(size=(nq, 3)).astype(np.float32)
    qR = rng.random((nq, 3)).astype(np.float32)
    r2 = np.where(rng.random(nq) < 0.5, 0.08, 0.8).astype(np.float32)
    of, ofound, or2, oidx = om.gather(qp, qn, qwo, qR, r2)
    for mode in (1, 0):
        t0 = time.time()
        gf, gfound, gr2, gidx = gm.gather(qp, qn, qwo, qR, r2, mode=mode)
        dt = time.time() - t0
        rel = np.abs(gf - of) / np.maximum(np.abs(of), 1e-20)
        print("gather mode %d: %.3fs found mismatch %d r2 mismatch %d idx mismatch rows %d flux maxrel %.3g (found mean %.1f)" % (
            mode, dt, (gfound != ofound).sum(), (gr2 != or2).sum(), (gidx != oidx).any(axis=1).sum(),
            rel[np.abs(of) > 1e-12].max() if (np.abs(of) > 1e-12).any() else 0, gfound.mean()), flush=True)

    # full iterations
    gs, cs = 20000, (5000 if scene.tracer.sppm.causticSize else 0)
    tr = SppmTracer.from_scene(scene, globalSize=gs, causticSize=cs, numRecurse=4, device=0)
    tr.setScene(scene)
    tr.setCamera(W, W)
    tr.startRaytracing()
    orr = O.OracleSppm(osc, scene.sppm_param(globalSize=gs, causticSize=cs, numRecurse=4), scene.tracer.max_depth,
                       scene.tracer.russian_roulette, 0x5EED, threads=8, split_mode=1, faithful_bvh=0)
    for it in range(4):
        st = tr.iteration()
        ost = orr.iteration(it)
        d = st.as_dict()
        print("iter", it, {k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in d.items()}, flush=True)
        print("   oracle", {k2: v for k2, v in ost.as_dict().items() if not k2.startswith("ms")}, flush=True)
        if it == 0:
            gpos, gpow, gth, gph = tr.photons(0)
            opos, opow, oth, oph, _ = orr.photons(0, gs)
            nn = min(len(gpos), len(opos))
            dd = np.abs(gpos[:nn] - opos[:nn]).max(axis=1)
            print("   photons: n gpu %d oracle %d; identical-position fraction %.4f; first divergence at %s" % (
                len(gpos), len(opos), (dd < 1e-4).mean(), np.argmax(dd >= 1e-4) if (dd >= 1e-4).any() else None), flush=True)
    gpx = tr.pixels()
    opx = orr.pixels()
    print("pixel planes mean gpu", gpx.mean(0).round(4))
    print("pixel planes mean orc", opx.mean(0).round(4))
    gi = tr.resolve().astype(np.float32)
    oi = orr.resolve().astype(np.float32)
    print("image rmse (0..255)", np.sqrt(((gi - oi) ** 2).mean()), "mean gpu", gi.mean(), "oracle", oi.mean(), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    tr.saveImage("gpurun_out/probe_%s.ppm" % name.split(".")[0])


if __name__ == "__main__":
    main()
