"""The benchmark configurations under test at full size: the three Cornell scenes as their scene files
configure them (whole 16-iteration renders), cornell_dragon with 2 M global + 500 k caustic photons per
iteration (split eight ways), and the synthetic height field (10 M triangles, 4096^2, 16 M + 4 M
photons).  The large two also at reduced size against the CPU oracle; full size goes through properties
that do not depend on the size."""
import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd import _abi
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import Context, PhotonMap, SppmTracer
from tests.test_parity_gpu import _assert_pixels_equal, _tracers

pytestmark = pytest.mark.gpu


def _pixel_invariants(px, k_g, k_c, r0, iterations):
    """What holds for every SPPMPixel array after `iterations` iterations, whatever the scene."""
    # A camera path that meets glass at exactly grazing incidence gets the reference's R / 0 from sample_f and
    # a NaN throughput, and with it NaN sums for its pixel (tests/test_parity_gpu.py::
    # test_non_finite_throughput_matches_oracle): a few pixels per million and render at most.
    bad = ~np.isfinite(px).all(axis=1)
    assert bad.sum() <= 4 * iterations and np.isfinite(px[:, 9:13]).all()
    px = px[~bad]
    assert (px[:, 0:9] >= 0).all()                                     # Ld and both fluxes
    cN, gN, cR, gR = px[:, 9], px[:, 10], px[:, 11], px[:, 12]
    assert (cR <= np.float32(r0)).all() and (gR <= np.float32(r0)).all() and (cR > 0).all() and (gR > 0).all()
    if iterations == 1:
        # first touch: N = found <= k; the radius shrinks exactly when k photons were found
        assert (gN <= k_g).all() and (cN <= k_c).all()
        assert ((gN == k_g) | (gR == np.float32(r0)))[gN > 0].all()
        assert ((cN == k_c) | (cR == np.float32(r0)))[cN > 0].all()
    else:
        assert (gN <= k_g * (1 + 0.7 * (iterations - 1)) + 1e-3).all()
    untouched = (gN == 0) & (cN == 0)
    assert (px[untouched, 3:9] == 0).all()


def _knn_properties(ctx, pos, power, th, ph, k, rng, nq):
    """Found counts, returned radii and index sets of a gather on a full-size map: the size-independent
    properties, both gather kernels against each other, and exhaustive search on a sample."""
    n = len(pos)
    m = PhotonMap(ctx, n, k)
    m.store(pos, power, th, ph)
    m.build()
    qp = (pos[rng.integers(0, n, nq)] + rng.normal(scale=0.01, size=(nq, 3))).astype(np.float32)
    qn = np.tile(np.array([[0, 1, 0]], np.float32), (nq, 1))
    qwo = rng.normal(size=(nq, 3)).astype(np.float32)
    qR = rng.random((nq, 3)).astype(np.float32)
    r2 = rng.choice(np.array([0.08, 0.002, 0.0002], np.float32), nq).astype(np.float32)
    flux, found, r2o, idx = m.gather(qp, qn, qwo, qR, r2, mode=0)
    assert found.min() >= 0 and found.max() <= k
    assert ((found < k) == (r2o == r2)).all() and (r2o <= r2).all()
    assert ((idx >= 0).sum(axis=1) == found).all()
    assert (flux >= 0).all() and np.isfinite(flux).all()
    assert (found == k).any() and (found < k).any()
    s = slice(0, min(nq, 20000))
    f1, n1, r1, i1 = m.gather(qp[s], qn[s], qwo[s], qR[s], r2[s], mode=1)
    assert (n1 == found[s]).all() and (i1 == idx[s]).all() and (r1 == r2o[s]).all()
    np.testing.assert_allclose(f1, flux[s], rtol=1e-5, atol=1e-10)
    for q in rng.integers(0, nq, 16):
        dv = pos - qp[q]
        d2 = (dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1]) + dv[:, 2] * dv[:, 2]
        cand = np.nonzero(d2 < r2[q])[0]
        order = cand[np.lexsort((cand, d2[cand]))][:k]
        assert sorted(order.tolist()) == idx[q][idx[q] >= 0].tolist()
    m.close()


# ------------------------------------------------- the three Cornell scenes as configured
@pytest.mark.parametrize("name,size", [("cornell_box.scene", 512), ("cornell_sphere.scene", 512),
                                       ("cornell_sphere.scene", 1024), ("cornell_dragon.scene", 1024)])
def test_cornell_scene_full_size_render(name, size):
    """A whole render of the scene file's own configuration (200 k global photons, 50 k caustic ones, none in
    the box scene, k = 100 / 50, 16 iterations) at the benchmark's image size: counters, pixel invariants after
    the first and the last iteration, radii that only shrink, and an image."""
    scene = Scene.load(asset_scene(name))
    prm = scene.tracer.sppm
    assert prm.globalSize == 200000 and prm.causticSize == (0 if "box" in name else 50000) and prm.numRecurse == 16
    tr = SppmTracer.from_scene(scene)
    tr.setScene(scene)
    tr.setCamera(size, size)
    tr.startRaytracing()
    P = size * size
    radii = None
    for it in range(16):
        st = tr.iteration()
        assert st.global_stored == 200000 and st.caustic_stored == prm.causticSize
        assert 0 < st.global_paths < 200000 and st.shoot_global >= (it + 1) * 1000
        assert (st.caustic_paths > prm.causticSize) if prm.causticSize else (st.caustic_paths == 0 and st.gathers_caustic == 0)
        assert 0.3 * P < st.gathers_global <= P and st.gathers_caustic <= P
        assert st.gather_photons <= 100 * st.gathers_global + 50 * st.gathers_caustic
        assert st.rays >= P + st.global_paths       # (caustic light paths that pass by everything specular are not traced)
        if it in (0, 15):
            px = tr.pixels()
            _pixel_invariants(px, 100, 50, 0.08, it + 1)
            if radii is not None:
                assert (px[:, 11] <= radii[0]).all() and (px[:, 12] <= radii[1]).all()
                assert (px[:, 12] < radii[1]).mean() > 0.5          # progressive: most radii have shrunk since iteration 0
            radii = (px[:, 11].copy(), px[:, 12].copy())
    img = tr.resolve()
    assert img.shape == (size, size, 3) and 20 < img.mean() < 235
    tr.close()


# ------------------------------------------------- cornell_dragon, 2 M / 500 k photons
def test_dragon_2m_full_size_iteration():
    """cornell_dragon_2m.scene as it is benchmarked: 1024^2, 2 M global + 500 k caustic photons, two iterations."""
    scene = Scene.load(asset_scene("cornell_dragon_2m.scene"))
    assert scene.tracer.sppm.globalSize == 2000000 and scene.tracer.sppm.causticSize == 500000
    tr = SppmTracer.from_scene(scene)
    tr.setScene(scene)
    tr.setCamera()
    tr.startRaytracing()
    P = tr.width * tr.height
    assert P == 1024 * 1024
    for it in range(2):
        st = tr.iteration()
        assert st.global_stored == 2000000 and st.caustic_stored == 500000
        assert 0 < st.global_paths < 2000000 and st.caustic_paths > 500000       # several photons per global path, few per caustic path
        assert st.shoot_global >= st.global_paths and st.shoot_caustic >= st.caustic_paths
        assert 0.5 * P < st.gathers_global <= P and 0 < st.gathers_caustic <= P
        assert st.gather_photons <= 100 * st.gathers_global + 50 * st.gathers_caustic
        assert st.rays >= P + st.global_paths + st.caustic_paths
        _pixel_invariants(tr.pixels(), 100, 50, 0.08, it + 1)
    img = tr.resolve()
    assert img.shape == (1024, 1024, 3) and 20 < img.mean() < 235
    # the photon maps of that iteration under the kNN properties
    rng = np.random.default_rng(11)
    for which, k in ((0, 100), (1, 50)):
        pos, power, th, ph = tr.photons(which)
        assert len(pos) == (2000000, 500000)[which]
        assert (np.abs(pos) < 5).all() and (power >= 0).all()
        _knn_properties(tr.ctx, pos, power, th, ph, k, rng, 200000 if which == 0 else 50000)
    tr.close()


def test_dragon_split_eight_ways_matches_oracle():
    """The eight-rank split of the dragon configuration, every rank a context of its own on this GPU, the photon
    exchange by hand, at reduced size against the oracle's ranks: the photon shards and the pixel state of every
    rank's rows are equal (fluxes to 1e-5)."""
    import torch
    from pathtracer_amd import distributed as D
    world = 8
    scene = Scene.load(asset_scene("cornell_dragon_2m.scene"))
    W, GS, CS = 64, 16000, 2400
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    bes = [D.HipBackend(scene, prm, 20, 0.8, 0x5EED, W, W, r, world, 0) for r in range(world)]
    osc = O.OracleScene(scene, W, W)
    ors = [O.OracleSppm(osc, prm, 20, 0.8, 0x5EED, rank=r, world=world, threads=4) for r in range(world)]
    for o in ors:
        o.set_tie_rule(True)
    for it in range(2):
        for be, orc in zip(bes, ors):
            be.photon_begin(it)
            be.camera_begin(it)
            be.photon_pass(it)
            orc.photon_pass(it)
        for which, size in ((0, GS), (1, CS)):
            parts = [be.export_local(which) for be in bes]
            assert [p[3] for p in parts] == [D.quota(size, r, world) for r in range(world)]
            cat = [torch.cat([p[i][:p[3]] for p in parts]).contiguous() for i in range(3)]
            shot = sum(p[4] for p in parts)
            for be in bes:
                be.set_map(which, cat[0], cat[1], cat[2], size, shot)
            op = [o.photons(which, size) for o in ors]
            assert shot == sum(p[4] for p in op)
            opos, opow = np.concatenate([p[0] for p in op]), np.concatenate([p[1] for p in op])
            oth, oph = np.concatenate([p[2] for p in op]), np.concatenate([p[3] for p in op])
            assert np.array_equal(cat[0].cpu().numpy().reshape(-1, 3), opos) and np.array_equal(cat[1].cpu().numpy().reshape(-1, 3), opow)
            assert np.array_equal(cat[2].cpu().numpy().view(np.uint16), oth.astype(np.uint16) | (oph.astype(np.uint16) << 8))
            for o in ors:
                o.set_map_photons(which, opos, opow, oth, oph, shot)
        for be, orc in zip(bes, ors):
            be.build_maps(); be.camera_pass(it)
            orc.build_maps(); orc.camera_pass(it)
    owner = np.array([D.row_owner(y, world) for y in range(W)])
    for r in range(world):
        mine = np.repeat(owner == r, W)
        gpx = bes[r].export_planes().cpu().numpy().reshape(13, W * W).T
        opx = ors[r].pixels()
        _assert_pixels_equal(np.ascontiguousarray(gpx[mine]), np.ascontiguousarray(opx[mine]))
        assert (gpx[~mine, :11] == 0).all() and (gpx[~mine, 11:] == np.float32(0.08)).all()
    for be in bes:
        be.close()
    for o in ors:
        o.close()
    osc.close()


# ------------------------------------------------------------- synthetic height field
def test_synthetic_field_reduced_matches_oracle():
    """The generator of the 10 M-triangle configuration at 200 x 200 quads, 256^2, 200 k + 50 k photons: ray hits,
    photon arrays, counters and pixels against the oracle."""
    scene = Scene.synthetic(200)
    assert scene.n_triangles == 2 * 200 * 200 + 10 and scene.n_spheres == 2
    W = 256
    tr, osc, orr = _tracers(scene, W, 200000, 50000, 4)
    rng = np.random.default_rng(5)
    n = 30000
    org = np.stack([rng.uniform(-2.1, 2.1, n), rng.uniform(-1.9, 2.1, n), rng.uniform(0.1, 4.3, n)], 1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 2, 1] = -np.abs(d[: n // 2, 1])                           # half of them towards the field
    gp, gt, gpp, gn, ge = tr.ctx.trace_rays(org, d)
    op, ot, opp, on, oe = osc.trace(org, d)
    assert (gp == op).all() and (gp >= 0).mean() > 0.7      # the shell has no ceiling
    for a, b in ((gt, ot), (gpp, opp), (gn, on), (ge, oe)):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
    for it in range(2):
        st = tr.iteration()
        ost = orr.iteration(it)
        for f in ("global_stored", "caustic_stored", "global_paths", "caustic_paths", "shoot_global", "shoot_caustic",
                  "gathers_global", "gathers_caustic", "gather_photons"):
            assert getattr(st, f) == getattr(ost, f), (it, f, getattr(st, f), getattr(ost, f))
        if it == 0:
            for which, cap in ((0, 200000), (1, 50000)):
                g4, o4 = tr.photons(which), orr.photons(which, cap)
                for a, b in zip(g4, o4[:4]):
                    assert np.array_equal(a, b)
    _assert_pixels_equal(tr.pixels(), orr.pixels())
    gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
    assert np.abs(gi - oi).max() <= 1
    tr.close(); orr.close(); osc.close()


def test_synthetic_field_full_size_iteration():
    """The configuration itself: 2236 x 2236 quads (9 999 402 triangles), 4096^2, 16 M global + 4 M caustic photons:
    one iteration, counters and pixel invariants, then the kNN properties on the 16 M-photon map."""
    scene = Scene.synthetic(2236)
    assert scene.n_triangles == 9999402
    prm = dict(globalSize=16000000, causticSize=4000000, numRecurse=64)
    tr = SppmTracer.from_scene(scene, **prm)
    tr.setScene(scene)
    tr.setCamera(4096, 4096)
    tr.startRaytracing()
    P = 4096 * 4096
    st = tr.iteration()
    assert st.global_stored == 16000000 and st.caustic_stored == 4000000
    assert 0 < st.global_paths < 16000000 and st.caustic_paths > 4000000
    assert 0.5 * P < st.gathers_global <= P and 0 < st.gathers_caustic <= P
    assert st.gather_photons <= 100 * st.gathers_global + 50 * st.gathers_caustic
    px = tr.pixels()
    _pixel_invariants(px, 100, 50, 0.08, 1)
    # at this photon density nearly every gather fills up and its radius shrinks
    seen = px[:, 10] > 0
    assert (px[seen, 10] == 100).mean() > 0.95 and (px[seen, 12] < np.float32(0.08)).mean() > 0.95
    del px
    img = tr.resolve()
    assert img.shape == (4096, 4096, 3) and 20 < img.mean() < 235
    pos, power, th, ph = tr.photons(0)
    assert len(pos) == 16000000
    _knn_properties(tr.ctx, pos, power, th, ph, 100, np.random.default_rng(2), 100000)
    tr.close()
