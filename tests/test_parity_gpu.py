"""GPU parity tests: every stage of the HIP path against the CPU oracle on the
same seeded inputs and against the committed golden vectors, all through the
C ABI (libtkhip.so).  Bars: bit-exact for indices, counts, radii and ray hits;
1e-5 relative for fp32 flux sums (summation order differs: stated in each test).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd import _abi
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import Context, PhotonMap, SppmTracer
from tests.conftest import golden, make_photons, make_queries

pytestmark = pytest.mark.gpu
L = O.lib()
FLUX_RTOL = 1e-5


def _flux_close(a, b):
    scale = np.maximum(np.abs(b).max(axis=1, keepdims=True), 1e-20)
    assert (np.abs(a - b) <= FLUX_RTOL * scale + 1e-12).all(), np.abs((a - b) / scale).max()


# ------------------------------------------------------------------ scalar pieces
def test_radical_inverse_matches_golden_and_oracle(gpu_ctx):
    g = golden("radical_inverse.npz")
    assert (gpu_ctx.radical_inverse(g["dim"], g["index"]) == g["value"]).all()
    rng = np.random.default_rng(2)
    dim = rng.integers(0, 7, 20000).astype(np.int32)
    idx = rng.integers(0, 1 << 34, 20000).astype(np.uint64)
    o = np.array([L.orc_radical_inverse(int(d), int(i)) for d, i in zip(dim, idx)], np.float32)
    assert (gpu_ctx.radical_inverse(dim, idx) == o).all()


def test_radical_inverse_at_the_edges_of_its_digit_counts(gpu_ctx):
    """Indices around every power of the base (where the number of digits changes), around 2^32 (where the index
    leaves 32-bit arithmetic) and far beyond give the float of the oracle's digit loop."""
    cases = []
    for dim, base in ((1, 3), (2, 5), (3, 7), (4, 11)):
        p = 1
        while p < (1 << 63):
            for v in (p - 2, p - 1, p, p + 1, 2 * p - 1, 2 * p, p * (base - 1), p * base - 1):
                if 0 <= v < (1 << 64):
                    cases.append((dim, v))
            p *= base
        for v in list(range(0, 300)) + [(1 << 32) + d for d in range(-3, 4)] + [(1 << 64) - 1, (1 << 63) + 12345, 10 ** 18 + 7]:
            cases.append((dim, v))
    rng = np.random.default_rng(5)
    for dim in (1, 2, 3, 4):
        for bits in (20, 31, 33, 40, 52, 64):
            cases += [(dim, int(v)) for v in rng.integers(0, (1 << bits) - 1, 400, dtype=np.uint64, endpoint=True)]
    dim = np.array([c[0] for c in cases], np.int32)
    idx = np.array([c[1] for c in cases], np.uint64)
    o = np.array([L.orc_radical_inverse(int(d), int(i)) for d, i in zip(dim, idx)], np.float32)
    g = gpu_ctx.radical_inverse(dim, idx)
    assert (g == o).all(), [(int(d), int(i)) for d, i in zip(dim[g != o][:5], idx[g != o][:5])]


def test_pixel_update_bit_exact(gpu_ctx):
    g = golden("pixel_update.npz")
    f, p, r = gpu_ctx.pixel_update(g["flux"], g["photons"], g["radius2"], g["lindir"], g["found"], g["r2ret"])
    assert (f == g["flux_out"]).all() and (p == g["photons_out"]).all() and (r == g["radius2_out"]).all()


# ------------------------------------------------------------------ materials and light emission
def test_sample_f_matches_golden(gpu_ctx):
    """Material::sample_f of the three materials in both transport modes.  Tolerance: 2e-6 absolute on
    directions and 1e-5 relative on f / pdf (the disk mapping calls sinf / cosf, whose last bit may differ
    between the device and the host libm); the reflect / refract branches themselves must agree."""
    g = golden("sample_f.npz")
    wo, nrm, u = g["wo"], g["nrm"], g["u"]
    for name in ("diffuse", "mirror", "glass"):
        m = _abi.TkMaterial.from_buffer_copy(bytes(g[name + "_material"]))
        for mode in (0, 1):
            wi, f, pdf = gpu_ctx.sample_f(m, mode, wo, nrm, u)
            ewi, ef, epdf = g[f"{name}_wi_{mode}"], g[f"{name}_f_{mode}"], g[f"{name}_pdf_{mode}"]
            assert ((pdf == 0) == (epdf == 0)).all(), (name, mode)
            assert np.abs(wi - ewi).max() < 2e-6, (name, mode)
            assert np.allclose(pdf, epdf, rtol=1e-5, atol=1e-7), (name, mode)
            assert np.allclose(f, ef, rtol=1e-5, atol=1e-7), (name, mode)
            if name != "diffuse":
                assert np.array_equal(wi, ewi) and np.array_equal(pdf, epdf)      # no transcendental on these paths


def test_sample_le_matches_golden(gpu_ctx, sphere_scene):
    """AreaLight::sample_Le: point and pdfs bit-exact, the cosine-distributed direction to the last bit of
    sinf / cosf (2e-6), the spawned origin to one ulp of the offset."""
    g = golden("sample_le_cornell_sphere.npz")
    gpu_ctx.set_scene(sphere_scene)
    org, d, nl, le, pp, pd = gpu_ctx.sample_le(0, g["u1"], g["u2"], g["upick"])
    assert np.array_equal(nl, g["nrm"]) and np.array_equal(le, g["le"]) and np.array_equal(pp, g["pdf_pos"])
    assert np.abs(d - g["dir"]).max() < 2e-6
    assert np.allclose(pd, g["pdf_dir"], rtol=1e-5, atol=1e-7)
    assert np.abs(org - g["org"]).max() < 2e-6
    with pytest.raises(_abi.TkError):
        gpu_ctx.sample_le(5, g["u1"], g["u2"], g["upick"])


# ------------------------------------------------------------------ rays
@pytest.mark.parametrize("name", ["cornell_sphere", "cornell_box"])
def test_ray_hits_match_golden(gpu_ctx, name):
    sc = Scene.load(asset_scene(name + ".scene"))
    g = golden(f"rays_{name}.npz")
    gpu_ctx.set_scene(sc)
    gpu_ctx.set_camera(sc, 512, 512)
    org, d = gpu_ctx.camera_rays(g["raster"])
    assert (org == g["org"]).all() and (d == g["dir"]).all()
    for o_, d_, keys in ((g["org"], g["dir"], ("prim", "t", "p", "nrm", "perr")),
                         (g["org2"], g["dir2"], ("prim2", "t2", "p2", "nrm2", "perr2"))):
        out = gpu_ctx.trace_rays(o_, d_)
        for got, key in zip(out, keys):
            assert (got == g[key]).all(), key


def test_ray_hits_match_oracle_on_the_dragon(gpu_ctx, dragon_scene):
    """100k triangles: primary rays, incoherent secondary rays and shadow-style rays with a t_max."""
    gpu_ctx.set_scene(dragon_scene)
    gpu_ctx.set_camera(dragon_scene, 1024, 1024)
    osc = O.OracleScene(dragon_scene, 1024, 1024)
    rng = np.random.default_rng(4)
    raster = np.stack([rng.random(60000) * 500 + 262, rng.random(60000) * 560 + 400], 1).astype(np.float32)  # around the dragon
    org, d = gpu_ctx.camera_rays(raster)
    oo, od = osc.camera_rays(raster)
    assert (org == oo).all() and (d == od).all()
    gp = gpu_ctx.trace_rays(org, d)
    op = osc.trace(org, d)
    for a, b in zip(gp, op):
        assert (a == b).all()
    hit = op[0] >= 0
    assert hit.mean() > 0.9 and (op[0][hit] >= 12).mean() > 0.3          # a good share of the hits is on the dragon
    m = int(hit.sum())
    d2 = rng.normal(size=(m, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = (op[2][hit] + 1e-3 * op[3][hit] * np.sign((d2 * op[3][hit]).sum(1, keepdims=True))).astype(np.float32)
    tmax = np.where(rng.random(m) < 0.5, 0.5, 3.0e38).astype(np.float32)
    gp2 = gpu_ctx.trace_rays(o2, d2, tmax)
    op2 = osc.trace(o2, d2, tmax)
    for a, b in zip(gp2, op2):
        assert (a == b).all()
    # the reference's traversal order (far child first, no culling) finds the same hits
    of = osc.trace(o2[:3000], d2[:3000], tmax[:3000], faithful=1)
    assert (of[0] == op2[0][:3000]).all() and (of[1] == op2[1][:3000]).all()
    osc.close()


# ------------------------------------------------------------------ kNN gather
def _gpu_map(ctx, g_or_arrays, k, tree=0):
    pos, power, th, ph = g_or_arrays
    m = PhotonMap(ctx, len(pos), k)
    m.set_tree(tree)
    m.store(pos, power, th, ph)
    m.build()
    return m


@pytest.mark.parametrize("name", ["knn_surface.npz", "knn_small_k50.npz", "knn_ties.npz"])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tree", [0, 1])
def test_knn_matches_golden(gpu_ctx, name, mode, tree):
    """Both gather kernels on both tree layouts (Hilbert order, median splits): the same neighbour sets."""
    g = golden(name)
    m = _gpu_map(gpu_ctx, (g["pos"], g["power"], g["theta"], g["phi"]), int(g["k"]), tree)
    flux, found, r2o, idx = m.gather(g["qp"], g["qn"], g["qwo"], g["qR"], g["r2"], mode=mode)
    assert (found == g["found"]).all()
    assert (idx == g["idx"]).all()            # neighbour index sets, ties broken by photon index
    assert (r2o == g["r2out"]).all()
    _flux_close(flux, g["flux"])
    m.close()


@pytest.mark.parametrize("n,k,nq", [(50000, 100, 20000), (9000, 50, 8000), (700, 448, 500), (37, 100, 300), (1, 5, 50)])
def test_knn_matches_oracle_random(gpu_ctx, n, k, nq):
    rng = np.random.default_rng(n + k)
    pos, power, th, ph = make_photons(rng, n)
    qp, qn, qwo, qR, r2 = make_queries(rng, nq, radius2=(0.08, 0.5, 1e-3, 30.0))
    om = O.OracleMap(n, k)
    om.store(pos, power, th, ph)
    om.build(1)
    of, ofound, or2, oidx = om.gather(qp, qn, qwo, qR, r2)
    for tree in (0, 1):
        gm = _gpu_map(gpu_ctx, (pos, power, th, ph), k, tree)
        nl, cap, depth, perm = gm.tree_info(n)
        assert sorted(perm.tolist()) == list(range(n)) and nl * cap >= n and nl == 1 << depth
        if tree == 1 and depth > 0:
            # median split of the root on the widest axis: the left half lies below the right half on it
            axis = int(np.argmax(pos.max(0) - pos.min(0)))
            half = (nl // 2) * n // nl
            assert pos[perm[:half], axis].max() <= pos[perm[half:], axis].min()
        for mode in (0, 1):
            gf, gfound, gr2, gidx = gm.gather(qp, qn, qwo, qR, r2, mode=mode)
            assert (gfound == ofound).all() and (gidx == oidx).all() and (gr2 == or2).all()
            _flux_close(gf, of)
        gm.close()
    om.close()


def test_knn_many_photons_at_one_distance(gpu_ctx):
    """Hundreds of photons at the same spot: the k-th nearest falls among keys that differ in the photon index only
    (more of them than the binned selection ranks at once).  Expected sets by the index rule (brute force): the CPU
    code keeps whichever of the tied photons its traversal meets first."""
    rng = np.random.default_rng(77)
    pos, power, th, ph = make_photons(rng, 3000)
    spots = pos[:3].copy()
    pos[100:500] = spots[0]
    pos[900:1100] = spots[1]
    pos[2000:2090] = spots[2]
    qp, qn, qwo, qR, r2 = make_queries(rng, 600, radius2=(0.08, 0.5, 30.0))
    qp[:300] = spots[rng.integers(0, 3, 300)] + rng.normal(scale=1e-3, size=(300, 3)).astype(np.float32)
    d = pos[None, :, :] - qp[:, None, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]          # the fp32 expression of the kernels
    for k in (50, 100, 448):
        want_idx = np.full((len(qp), k), 1 << 30, np.int64)
        want_found = np.zeros(len(qp), np.int32)
        want_r2 = r2.copy()
        for i in range(len(qp)):
            cand = np.nonzero(d2[i] < r2[i])[0]
            keep = cand[np.lexsort((cand, d2[i][cand]))[:k]]
            want_idx[i, :len(keep)] = np.sort(keep)
            want_found[i] = len(keep)
            if len(keep) == k:
                want_r2[i] = d2[i][keep].max()
        flux_ref = None
        for tree in (0, 1):
            gm = _gpu_map(gpu_ctx, (pos, power, th, ph), k, tree)
            for mode in (0, 1):
                gf, gfound, gr2, gidx = gm.gather(qp, qn, qwo, qR, r2, mode=mode)
                got = np.sort(np.where(gidx < 0, 1 << 30, gidx).astype(np.int64), axis=1)
                assert (gfound == want_found).all() and (got == want_idx).all() and (gr2 == want_r2).all()
                if flux_ref is None:
                    flux_ref = gf
                _flux_close(gf, flux_ref)      # same photons kept: the four ways of getting there agree
            gm.close()


def test_photon_begin_state_checks():
    """tkhip_sppm_photon_begin: a second call, or a whole iteration, while a pass is in flight is refused; the pass
    it starts is the one tkhip_sppm_photon_pass finishes."""
    from pathtracer_amd import distributed as D
    scene = Scene.load(asset_scene("cornell_box.scene"))
    prm = scene.sppm_param(globalSize=4000, causticSize=0, numRecurse=4)
    be = D.HipBackend(scene, prm, 20, 0.8, 0x5EED, 32, 32, 0, 1, 0)
    be.photon_begin(0)
    with pytest.raises(_abi.TkError):
        be.photon_begin(0)
    with pytest.raises(_abi.TkError):
        be.iteration_single(0)
    st = _abi.TkIterStats()
    be.photon_pass(0, st)
    assert st.global_stored == 4000 and st.global_paths > 0
    be.build_maps(); be.camera_pass(0)
    be.iteration_single(1)
    be.close()


def test_tiny_photon_maps_render():
    """Photon maps far smaller than any staging or scratch granule (100 + 10 photons, k larger than the maps)."""
    from pathtracer_amd import distributed as D
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    prm = scene.sppm_param(globalSize=100, causticSize=10, numRecurse=4)
    be = D.HipBackend(scene, prm, 20, 0.8, 0x5EED, 24, 24, 0, 1, 0)
    for it in range(3):
        st = _abi.TkIterStats()
        be.iteration_single(it, st)
        assert st.global_stored == 100 and st.caustic_stored == 10 and st.gathers_global > 0
    planes = be.export_planes().cpu().numpy()
    assert np.isfinite(planes).all()
    be.close()


def test_knn_edge_cases(gpu_ctx):
    rng = np.random.default_rng(8)
    qp, qn, qwo, qR, r2 = make_queries(rng, 130)
    # empty map: nothing found, radius unchanged
    m = PhotonMap(gpu_ctx, 16, 10)
    m.build()
    flux, found, r2o, idx = m.gather(qp, qn, qwo, qR, r2)
    assert (found == 0).all() and (flux == 0).all() and (r2o == r2).all() and (idx == -1).all()
    # zero and negative radii find nothing; gather before build is refused
    pos, power, th, ph = make_photons(rng, 500)
    m.close()
    m = PhotonMap(gpu_ctx, 500, 10)
    m.store(pos, power, th, ph)
    with pytest.raises(_abi.TkError):
        m.gather(qp, qn, qwo, qR, r2)
    m.build()
    r0 = np.zeros_like(r2)
    r0[::2] = -1
    flux, found, r2o, idx = m.gather(qp, qn, qwo, qR, r0)
    assert (found == 0).all() and (r2o == r0).all()
    # capacity and k limits
    with pytest.raises(_abi.TkError):
        m.store(pos, power, th, ph)
    with pytest.raises(_abi.TkError):
        PhotonMap(gpu_ctx, 10, _abi.TKHIP_MAX_K + 1)
    m.close()


def test_knn_full_size_properties(gpu_ctx):
    """200k photons, 1M queries (the benchmark's map size): size-independent properties + a brute-force sample."""
    rng = np.random.default_rng(77)
    n, k, nq = 200000, 100, 1 << 20
    pos, power, th, ph = make_photons(rng, n)
    qp, qn, qwo, qR, r2 = make_queries(rng, nq, radius2=(0.02, 0.08, 0.002))
    gm = _gpu_map(gpu_ctx, (pos, power, th, ph), k)
    flux, found, r2o, idx = gm.gather(qp, qn, qwo, qR, r2, mode=0)
    assert found.min() >= 0 and found.max() <= k
    assert ((found < k) == (r2o == r2)).all() and (r2o <= r2).all()
    assert ((idx >= 0).sum(axis=1) == found).all()
    assert (flux >= 0).all() and np.isfinite(flux).all()
    # mode 1 (per-query kernel only) agrees on a slice
    f1, n1, r1, i1 = gm.gather(qp[:50000], qn[:50000], qwo[:50000], qR[:50000], r2[:50000], mode=1)
    assert (n1 == found[:50000]).all() and (i1 == idx[:50000]).all() and (r1 == r2o[:50000]).all()
    _flux_close(f1, flux[:50000])
    # brute force on a sample
    for q in rng.integers(0, nq, 40):
        dv = pos - qp[q]
        d2 = (dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1]) + dv[:, 2] * dv[:, 2]
        cand = np.nonzero(d2 < r2[q])[0]
        order = cand[np.lexsort((cand, d2[cand]))][:k]
        assert sorted(order.tolist()) == idx[q][idx[q] >= 0].tolist()
    gm.close()


# ------------------------------------------------------------------ whole passes
def _tracers(scene, W, gs, cs, iters, seed=0x5EED):
    tr = SppmTracer.from_scene(scene, globalSize=gs, causticSize=cs, numRecurse=iters, device=0, seed=seed)
    tr.setScene(scene)
    tr.setCamera(W, W)
    tr.startRaytracing()
    osc = O.OracleScene(scene, W, W)
    orr = O.OracleSppm(osc, scene.sppm_param(globalSize=gs, causticSize=cs, numRecurse=iters), scene.tracer.max_depth,
                       scene.tracer.russian_roulette, seed, threads=8, split_mode=1, faithful_bvh=0)
    orr.set_tie_rule(True)   # photons tied at the k-th distance: lowest index, as on the GPU
    return tr, osc, orr


def _assert_pixels_equal(gpx, opx):
    """SPPMPixel arrays of the GPU and of the oracle after the same iterations: Ld, photon counts
    and radii bit for bit; the two fluxes within 1e-5 relative (the order of the sum over a gather's
    photons differs), with an absolute floor of 1e-7 of the largest flux for near-black pixels."""
    assert gpx.shape == opx.shape
    g, o = gpx.view(np.uint32), opx.view(np.uint32)
    # a throughput that went non-finite (a reflection at exactly grazing incidence, f = R / 0) turns a pixel's
    # sums NaN on both sides, in the same places; which NaN the hardware makes is not compared
    both_nan = np.isnan(gpx) & np.isnan(opx)
    assert ((g[:, 0:3] == o[:, 0:3]) | both_nan[:, 0:3]).all(), "Ld"
    assert ((g[:, 9:13] == o[:, 9:13]) | both_nan[:, 9:13]).all(), "photon counts / radii"
    for lo, hi in ((3, 6), (6, 9)):
        a, b = gpx[:, lo:hi], opx[:, lo:hi]
        with np.errstate(invalid="ignore"):
            close = np.abs(a - b) <= 1e-5 * np.abs(b) + 1e-7 * np.nanmax(np.abs(b))
        assert (close | both_nan[:, lo:hi]).all(), (lo, float(np.nanmax(np.abs(a - b))))


@pytest.mark.parametrize("name,cs", [("cornell_sphere.scene", 3000), ("cornell_box.scene", 0), ("cornell_dragon.scene", 3000)])
def test_photon_pass_matches_oracle(name, cs):
    """tracePhoton / traceCausticPhoton with the same Halton indices and the same hashed random numbers: the
    path counts and the photon arrays (positions, powers, direction bytes, in storage order) are equal bit
    for bit."""
    scene = Scene.load(asset_scene(name))
    tr, osc, orr = _tracers(scene, 64, 20000, cs, 4)
    st = _abi.TkIterStats()
    tr.photon_pass(0, st)
    ost = orr.photon_pass(0)
    assert st.global_stored == 20000 and st.caustic_stored == cs
    assert st.global_paths == ost.global_paths and st.caustic_paths == ost.caustic_paths
    assert st.shoot_global == ost.shoot_global and st.shoot_caustic == ost.shoot_caustic
    for which, cap in ((0, 20000), (1, cs)):
        if cap == 0:
            continue
        gpos, gpow, gth, gph = tr.photons(which)
        opos, opow, oth, oph, _ = orr.photons(which, cap)
        assert len(gpos) == len(opos) == cap
        assert (gpos.view(np.uint32) == opos.view(np.uint32)).all()
        assert (gpow.view(np.uint32) == opow.view(np.uint32)).all()
        assert (gth == oth).all() and (gph == oph).all()
    tr.close(); orr.close(); osc.close()


def test_encode_direction_matches_oracle(gpu_ctx):
    """The two direction bytes (float acos / atan2, float products) on random and on awkward directions."""
    rng = np.random.default_rng(3)
    wo = rng.normal(size=(50000, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wo[:6] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [-1, 0, 0], [0, -1, 0], [-1, -1e-8, 0]]
    wo[6] = [0, 0, np.float32(1.0000001)]                 # acos of a value just above one: NaN, clamped
    th, ph = gpu_ctx.encode_directions(wo)
    oth, oph = np.empty(len(wo), np.uint8), np.empty(len(wo), np.uint8)
    for i in range(len(wo)):
        a, b = C.c_uint8(), C.c_uint8()
        L.orc_encode_direction(O.fp(wo[i]), C.byref(a), C.byref(b))
        oth[i], oph[i] = a.value, b.value
    assert (th == oth).all() and (ph == oph).all()


@pytest.mark.parametrize("name,W,cs", [("cornell_sphere.scene", 96, 4000), ("cornell_dragon.scene", 128, 3000)])
def test_iterations_match_oracle(name, W, cs):
    """Three full iterations (photon passes, tree builds, camera pass with both gathers, update): counters
    equal, per-pixel SPPM state equal (fluxes to 1e-5), resolved images equal to one level."""
    scene = Scene.load(asset_scene(name))
    tr, osc, orr = _tracers(scene, W, 20000, cs, 4)
    for it in range(3):
        st = tr.iteration()
        ost = orr.iteration(it)
        for f in ("global_stored", "caustic_stored", "global_paths", "caustic_paths", "shoot_global", "shoot_caustic",
                  "gathers_global", "gathers_caustic", "gather_photons"):
            assert getattr(st, f) == getattr(ost, f), (it, f, getattr(st, f), getattr(ost, f))
        _assert_pixels_equal(tr.pixels(), orr.pixels())
    gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
    assert np.abs(gi - oi).max() <= 1 and (gi == oi).mean() > 0.999      # powf may differ in its last bit
    assert tr.counters() == orr.counters()
    tr.close(); orr.close(); osc.close()


@pytest.mark.parametrize("name,size", [("cornell_box.scene", 512), ("cornell_dragon.scene", 1024)])
def test_full_size_iterations_match_oracle(name, size):
    """The benchmark configurations themselves (image size and photon counts of the scene file) for two
    iterations: counters equal, every pixel's Ld, photon counts and radii bit for bit, fluxes to 1e-5."""
    scene = Scene.load(asset_scene(name))
    prm = scene.tracer.sppm
    tr, osc, orr = _tracers(scene, size, prm.globalSize, prm.causticSize, 16)
    for it in range(2):
        st, ost = tr.iteration(), orr.iteration(it)
        for f in ("global_paths", "caustic_paths", "global_stored", "caustic_stored", "gathers_global", "gathers_caustic"):
            assert getattr(st, f) == getattr(ost, f), (it, f, getattr(st, f), getattr(ost, f))
    _assert_pixels_equal(tr.pixels(), orr.pixels())
    tr.close(); orr.close(); osc.close()


def test_cornell_box_whole_render_matches_oracle():
    """The first benchmark configuration from end to end: cornell_box at 512^2, 200 k global photons, no caustic
    map, k = 100, all 16 iterations: the pixel state is compared after every fourth iteration and the resolved
    images at the end (1 LSB: powf may differ in its last bit)."""
    scene = Scene.load(asset_scene("cornell_box.scene"))
    prm = scene.tracer.sppm
    assert (scene.width, prm.globalSize, prm.causticSize, prm.numRecurse, prm.globalSample) == (512, 200000, 0, 16, 100)
    tr, osc, orr = _tracers(scene, 512, prm.globalSize, prm.causticSize, prm.numRecurse)
    for it in range(16):
        st, ost = tr.iteration(), orr.iteration(it)
        assert (st.global_paths, st.gathers_global, st.gathers_caustic) == (ost.global_paths, ost.gathers_global, 0)
        if it % 4 == 3:
            _assert_pixels_equal(tr.pixels(), orr.pixels())
    gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
    assert np.abs(gi - oi).max() <= 1 and (gi == oi).mean() > 0.999
    assert tr.counters() == orr.counters()
    tr.close(); orr.close(); osc.close()


def test_non_finite_throughput_matches_oracle():
    """cornell_sphere at 1024^2, full photon counts: in iteration 2 the camera path of pixel (499, 205) touches
    a sphere so that a specular sample_f divides by AbsDot(wi, N) = 0 (the reflection of the glass,
    src/core/Material.cpp:64-71, or of the mirror, :90-95, return R / 0): the throughput is inf * 0 = NaN from
    there on, `Ld += coef * black` at the next hit and the gathers behind it turn the pixel's Ld and both
    fluxes NaN (src/tracer/TkSppmTracer.cpp:315-318, :325-340).  The reference's arithmetic, reproduced: the same
    entries are NaN on both sides and every other pixel is equal as usual."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    tr, osc, orr = _tracers(scene, 1024, 200000, 50000, 16)
    for it in range(3):
        tr.iteration()
        orr.iteration(it)
    gpx, opx = tr.pixels(), orr.pixels()
    assert (np.isnan(gpx) == np.isnan(opx)).all()
    assert np.isnan(gpx[210419, 0:9]).all() and np.isfinite(gpx[210419, 9:13]).all()
    assert np.isnan(gpx).any(axis=1).sum() <= 4
    _assert_pixels_equal(gpx, opx)
    # the resolve clamps with tk::min / tk::max (src/TkMath.h:250-253), which hand a NaN on as the upper bound: white
    gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
    assert np.abs(gi - oi).max() <= 1
    assert (gi[1024 - 1 - 205, 499] == 255).all()
    tr.close(); orr.close(); osc.close()


def test_odd_image_size_matches_oracle():
    """Width and height that are multiples of nothing (the reference never traces its trailing tiles at such
    sizes; here every pixel is): two iterations against the oracle."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W, H = 53, 37
    prm = dict(globalSize=6000, causticSize=900, numRecurse=4)
    tr = SppmTracer.from_scene(scene, device=0, seed=7, **prm)
    tr.setScene(scene)
    tr.setCamera(W, H)
    tr.startRaytracing()
    osc = O.OracleScene(scene, W, H)
    orr = O.OracleSppm(osc, scene.sppm_param(**prm), scene.tracer.max_depth, scene.tracer.russian_roulette, 7, threads=4)
    for it in range(2):
        st = tr.iteration()
        ost = orr.iteration(it)
        assert st.global_stored == ost.global_stored == 6000
        assert st.gathers_global == ost.gathers_global and st.gathers_caustic == ost.gathers_caustic
    gpx, opx = tr.pixels(), orr.pixels()
    assert gpx.shape == opx.shape == (W * H, 13)
    assert (gpx[:, 12] <= np.float32(0.08)).all()
    _assert_pixels_equal(gpx, opx)
    gi, oi = tr.resolve().astype(int), orr.resolve().astype(int)
    assert gi.shape == (H, W, 3)
    assert np.abs(gi - oi).max() <= 1
    tr.close(); orr.close(); osc.close()


def test_gather_on_identical_maps_and_hitpoints():
    """The GPU's own photon maps and the oracle's camera hitpoints: with identical inputs the gather and the
    SPPMPixel update agree to the last bit in counts / radii and to 1e-5 in flux."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W = 64
    tr, osc, orr = _tracers(scene, W, 15000, 3000, 4)
    tr.iteration()
    ctx = tr.ctx
    for which, k in ((0, 100), (1, 50)):
        pos, power, th, ph = tr.photons(which)
        orr.set_map_photons(which, pos, power, th, ph, 1)
        gm = PhotonMap(ctx, len(pos), k)
        gm.store(pos, power, th, ph)
        gm.build()
        om = O.OracleMap(len(pos), k)
        om.store(pos, power, th, ph)
        om.build(0)                                       # reference tree shape (z-only splits)
        qs = []
        for y in range(0, W, 2):
            for x in range(0, W, 2):
                _, q = orr.camera_path(1, x, y)
                qs += [e for e in q if e.which == which]
        assert len(qs) > 200
        qp = np.array([list(q.p) for q in qs], np.float32); qn = np.array([list(q.n) for q in qs], np.float32)
        qwo = np.array([list(q.wo) for q in qs], np.float32); qR = np.array([list(q.R) for q in qs], np.float32)
        r2 = np.full(len(qs), 0.03, np.float32)
        of, ofound, or2, oidx = om.gather(qp, qn, qwo, qR, r2)
        gf, gfound, gr2, gidx = gm.gather(qp, qn, qwo, qR, r2)
        assert (gfound == ofound).all() and (gidx == oidx).all() and (gr2 == or2).all()
        _flux_close(gf, of)
        # the update rule on top of it
        coef = np.array([list(q.coef) for q in qs], np.float32)
        f0 = np.zeros((len(qs), 3), np.float32); p0 = np.zeros(len(qs), np.float32)
        out = gpu_update = ctx.pixel_update(f0, p0, r2, coef * of, ofound, or2)
        f1, p1, rr1 = f0.copy(), p0.copy(), r2.copy()
        L.orc_pixel_update(len(qs), O.fp(f1), O.fp(p1), O.fp(rr1), O.fp(np.ascontiguousarray(coef * of)), O.ip(ofound), O.fp(or2))
        assert (out[0] == f1).all() and (out[1] == p1).all() and (out[2] == rr1).all()
        gm.close(); om.close()
    tr.close(); orr.close(); osc.close()


def test_resolve_matches_oracle():
    scene = Scene.load(asset_scene("cornell_box.scene"))
    tr, osc, orr = _tracers(scene, 48, 5000, 0, 4)
    for _ in range(2):
        tr.iteration()
    px = tr.pixels()
    sg, sc_, nr = tr.counters()
    buf = (_abi.TkSPPMPixel * len(px)).from_buffer_copy(np.ascontiguousarray(px).tobytes())
    oimg = np.zeros((48, 48, 3), np.uint8)
    L.orc_resolve_rgb8(buf, 48, 48, nr, sg, sc_, 0, O.bp(oimg))
    gimg = tr.resolve()
    assert np.abs(gimg.astype(int) - oimg.astype(int)).max() <= 1      # powf may differ in its last bit
    assert (gimg == oimg).mean() > 0.99
    tr.close(); orr.close(); osc.close()


@pytest.mark.parametrize("world", [2, 3])
def test_rank_sharding_on_one_gpu(world):
    """Every rank of a world of 2 (3: quotas with a remainder) as a context of its own on this GPU, exchange by hand:
    the pixel state of every rank's rows equals the oracle's for the same sharding (fluxes to 1e-5)."""
    import torch
    from pathtracer_amd import distributed as D
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W, GS, CS = 64, 8000, 1000
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    bes = [D.HipBackend(scene, prm, 20, 0.8, 0x5EED, W, W, r, world, 0) for r in range(world)]
    osc = O.OracleScene(scene, W, W)
    ors = [O.OracleSppm(osc, prm, 20, 0.8, 0x5EED, rank=r, world=world, threads=4) for r in range(world)]
    for o in ors:
        o.set_tie_rule(True)
    for it in range(2):
        for be, orc in zip(bes, ors):
            be.photon_begin(it)        # light paths enqueued first, camera paths traced beside them,
            be.camera_begin(it)        # as the multi-rank driver does
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
            oshot = sum(p[4] for p in op)
            assert shot == oshot
            for o in ors:
                o.set_map_photons(which, np.concatenate([p[0] for p in op]), np.concatenate([p[1] for p in op]),
                                  np.concatenate([p[2] for p in op]), np.concatenate([p[3] for p in op]), oshot)
        for be, orc in zip(bes, ors):
            be.build_maps(); be.camera_pass(it)
            orc.build_maps(); orc.camera_pass(it)
    planes = [be.export_planes().cpu().numpy() for be in bes]
    oplanes = [o.pixels().T.reshape(13, W, W) for o in ors]
    owner = np.array([D.row_owner(y, world) for y in range(W)])
    for r in range(world):
        mine, other = owner == r, owner != r
        assert (planes[r][:11, other] == 0).all() and (planes[r][11:, other] == np.float32(0.08)).all()
        a, b = planes[r][:, mine].reshape(13, -1).T, oplanes[r][:, mine].reshape(13, -1).T
        _assert_pixels_equal(np.ascontiguousarray(a), np.ascontiguousarray(b))
    # import / export round trip and resolve on the merged state
    merged = planes[0].copy()
    for r in range(1, world):
        merged[:, owner == r] = planes[r][:, owner == r]
    bes[0].import_planes(torch.from_numpy(merged).cuda())
    assert np.array_equal(bes[0].export_planes().cpu().numpy(), merged)
    img = bes[0].resolve()
    assert img.shape == (W, W, 3) and img.mean() > 10
    for be in bes:
        be.close()
    for o in ors:
        o.close()
    osc.close()


def test_photon_exchange_buffer_on_one_gpu(monkeypatch):
    """The one-collective photon exchange of the multi-rank driver (one buffer per rank, laid out, filled and
    unpacked by the library on the device) against the exchange by hand, for a world of 3 whose quotas leave a
    remainder; the all-gather is played by handing every rank's buffer to rank 0."""
    import torch
    from pathtracer_amd import distributed as D
    world = 3
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W, GS, CS = 32, 8000, 1000
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    bes = [D.HipBackend(scene, prm, 20, 0.8, 0x5EED, W, W, r, world, 0) for r in range(world)]
    for be in bes:
        be.photon_pass(0)
    hand = []
    for which in (0, 1):
        parts = [be.export_local(which) for be in bes]
        hand.append(([torch.cat([p[i][:p[3]] for p in parts]).cpu().numpy() for i in range(3)], sum(p[4] for p in parts)))
    class Recorded(Exception):
        pass

    sent = {}
    for r, be in enumerate(bes):
        def record(t, w, r=r):
            torch.cuda.synchronize()
            sent[r] = t.clone()
            raise Recorded()
        monkeypatch.setattr(D, "_all_gather_flat", record)
        with pytest.raises(Recorded):
            D.exchange_photons(be, world)
    assert len({int(t.numel()) for t in sent.values()}) == 1        # equal sizes: one plain all-gather
    assert int(sent[0].numel()) == bes[0].exchange_words()
    monkeypatch.setattr(D, "_all_gather_flat", lambda t, w: torch.cat([sent[r] for r in range(w)]))
    before = bes[0].tracer.counters()
    D.exchange_photons(bes[0], world)
    bes[0].build_maps()               # settles the counts and shot totals the device summed
    after = bes[0].tracer.counters()
    for which, size in ((0, GS), (1, CS)):
        (pos, power, direction), shot = hand[which]
        assert after[which] - before[which] == shot
        gp, gw, th, ph = bes[0].tracer.photons(which)
        assert len(gp) == size
        assert np.array_equal(gp, pos) and np.array_equal(gw, power)
        assert np.array_equal(th.astype(np.uint16) | (ph.astype(np.uint16) << 8), direction.view(np.uint16))
    bes[0].camera_pass(0)
    assert bes[0].resolve().mean() > 1
    for be in bes:
        be.close()


def test_photon_exchange_with_a_quota_no_rank_can_fill(monkeypatch):
    """cornell_box holds nothing specular: no rank stores a caustic photon (its pass gives up at the emission
    cap), so the map comes out smaller than the sum of the quotas the host assumed when it enqueued the builds.
    The settled render equals the one whose exchange is made by hand with the true counts."""
    import torch
    from pathtracer_amd import distributed as D
    world = 2
    scene = Scene.load(asset_scene("cornell_box.scene"))
    W, GS, CS = 32, 4000, 40
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    dev = [D.HipBackend(scene, prm, 20, 0.8, 0x5EED, W, W, r, world, 0) for r in range(world)]
    hand = [D.HipBackend(scene, prm, 20, 0.8, 0x5EED, W, W, r, world, 0) for r in range(world)]
    for it in range(2):
        for be in dev + hand:
            be.photon_pass(it)
        sent = {}
        class Recorded(Exception):
            pass
        for r, be in enumerate(dev):
            def record(t, w, r=r):
                torch.cuda.synchronize()
                sent[r] = t.clone()
                raise Recorded()
            monkeypatch.setattr(D, "_all_gather_flat", record)
            with pytest.raises(Recorded):
                D.exchange_photons(be, world)
        monkeypatch.setattr(D, "_all_gather_flat", lambda t, w: torch.cat([sent[r] for r in range(w)]))
        for be in dev:
            D.exchange_photons(be, world)
        for which in (0, 1):
            parts = [be.export_local(which) for be in hand]
            if which == 1:
                assert [p[3] for p in parts] == [0, 0]
            cat = [torch.cat([p[i][:p[3]] for p in parts]).contiguous() for i in range(3)]
            for be in hand:
                be.set_map(which, cat[0], cat[1], cat[2], sum(p[3] for p in parts), sum(p[4] for p in parts))
        for be in dev + hand:
            be.build_maps()
            be.camera_pass(it)
    for a, b in zip(dev, hand):
        assert a.tracer.counters() == b.tracer.counters()
        assert np.array_equal(a.export_planes().cpu().numpy(), b.export_planes().cpu().numpy())
    for be in dev + hand:
        be.close()


def test_cpp_host_cli_renders_the_same_image(tmp_path):
    """The C++ host program (loader + SppmTracer class over the C ABI) against the Python-driven render."""
    scene_text = open(asset_scene("cornell_sphere.scene")).read()
    scene_text = scene_text.replace('<gsize i="200000"/>', '<gsize i="6000"/>').replace('<csize i="50000"/>', '<csize i="800"/>')
    scene_text = scene_text.replace('<recurse i="16"/>', '<recurse i="4"/>').replace("../models/", os.path.join(os.path.dirname(asset_scene("x")), "../models/"))
    sf = tmp_path / "small.scene"
    sf.write_text(scene_text)
    out = tmp_path / "cli.ppm"
    exe = os.path.join(_abi.LIB_DIR, "raytracer")
    res = subprocess.run([exe, str(sf), "--width", "80", "--height", "80", "--out", str(out)], capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr
    assert "Render complete" in res.stdout and "Rendering Done" in res.stderr
    data = out.read_bytes()
    assert data.startswith(b"P6\n80 80\n255\n")
    cli = np.frombuffer(data[len(b"P6\n80 80\n255\n"):], np.uint8).reshape(80, 80, 3)
    sc = Scene.load(str(sf))
    tr = SppmTracer.from_scene(sc, device=0)
    tr.setScene(sc)
    tr.setCamera(80, 80)
    img = tr.render()
    assert np.array_equal(cli, img)
    tr.close()
    # the timed, previewed steps of the windowed path render the same image as the headless loop
    out2 = tmp_path / "cli_steps.ppm"
    res = subprocess.run([exe, str(sf), "--width", "80", "--height", "80", "--out", str(out2)], capture_output=True, text=True,
                         timeout=300, env=dict(os.environ, TKHOST_ITER_STATS="1"))
    assert res.returncode == 0, res.stderr
    assert "Shoot global photons" in res.stderr
    assert out2.read_bytes() == data
