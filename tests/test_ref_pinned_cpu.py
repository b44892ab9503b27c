"""The CPU oracle against outputs of the upstream renderer itself.

tests/golden/ref_*.npz were written by oracle/ref_probe.cpp linked with
upstream's own translation units (oracle/build_ref.sh, oracle/make_ref_golden.py):
every array named *_value / *_out / hits / pixels below is what upstream's
function returned for the inputs stored beside it.  These tests pin the oracle;
tests/test_ref_pinned_gpu.py checks the HIP kernels against the same files.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd import _abi
from pathtracer_amd.scene import Scene, asset_scene
from tests.conftest import golden
from tests.ref_fixtures import (MATERIALS, TILE_DRAWS, assert_pixels_match, bits_equal, light_pick_candidates,
                                object_of_prim, sorted_idx, tk_material, ulp_distance)

L = O.lib()


def test_radical_inverse_hash_permutation():
    g = golden("ref_lowdiscrepancy.npz")
    v = np.array([L.orc_radical_inverse(int(d), int(i)) for d, i in zip(g["radinv_dim"], g["radinv_index"])], np.float32)
    assert bits_equal(v, g["radinv_value"])
    h = np.array([L.orc_hash_pixel(int(x), int(y), int(d), int(s))
                  for x, y, d, s in zip(g["hash_x"], g["hash_y"], g["hash_dim"], g["hash_seed"])], np.uint64)
    assert (h == g["hash_value"]).all()
    p = np.array([L.orc_permutation_element(int(i), int(l), int(q))
                  for i, l, q in zip(g["perm_i"], g["perm_l"], g["perm_p"])], np.int32)
    assert (p == g["perm_value"]).all()


def test_stratified_sampler_without_jitter():
    """(stratum + 0.5) / n from Hash + PermutationElement: the oracle's pieces reproduce upstream's sampler."""
    g = golden("ref_lowdiscrepancy.npz")
    for i in range(len(g["strat_x"])):
        x, y, idx, xs, ys = (int(g[k][i]) for k in ("strat_x", "strat_y", "strat_idx", "strat_xs", "strat_ys"))
        spp, dim, out = xs * ys, 0, []
        for d in TILE_DRAWS:
            st = L.orc_permutation_element(idx, spp, L.orc_hash_pixel(x, y, dim, 0) & 0xFFFFFFFF)
            if d == 1:
                out.append(np.float32(st + np.float32(0.5)) / np.float32(spp))
            else:
                out += [np.float32((st % xs) + np.float32(0.5)) / np.float32(xs),
                        np.float32((st // xs) + np.float32(0.5)) / np.float32(ys)]
            dim += d
        assert bits_equal(np.array(out, np.float32), g["strat_values"][i]), i


def test_direction_tables():
    g = golden("ref_knn.npz")
    t = [np.empty(256, np.float32) for _ in range(4)]
    L.orc_direction_tables(*[O.fp(a) for a in t])
    assert bits_equal(np.stack(t), g["dirtab"])


@pytest.mark.parametrize("tag", ["knn100", "knn50"])
@pytest.mark.parametrize("split,brute", [(0, False), (1, False), (1, True)])
def test_knn_sets_match_upstream(tag, split, brute):
    """neighbor_search + radiance_estimate: found, returned radius and index sets equal, flux to 1e-5."""
    g = golden("ref_knn.npz")
    k, n = int(g[tag + "_k"][0]), len(g[tag + "_theta"])
    m = O.OracleMap(n, k)
    m.store(g[tag + "_pos"], g[tag + "_power"], g[tag + "_theta"], g[tag + "_phi"])
    m.build(split)
    fl, fo, r2, idx = m.gather(g[tag + "_q_p"], g[tag + "_q_n"], g[tag + "_q_wo"], g[tag + "_q_R"], g[tag + "_q_radius2"],
                               brute=brute)
    m.close()
    assert (fo == g[tag + "_found"]).all()
    assert bits_equal(r2, g[tag + "_radius2_out"])
    assert (sorted_idx(idx) == g[tag + "_idx"]).all()
    np.testing.assert_allclose(fl, g[tag + "_flux"], rtol=1e-5, atol=1e-12)


def test_knn_ties_match_upstream_up_to_the_tie():
    """Photons tied at the k-th distance: upstream keeps whichever its traversal met first, the
    contract here is the lower index.  Count, radius, the distances of the set and (equal powers and
    directions in this fixture) the estimate are the same either way."""
    g = golden("ref_knn.npz")
    tag, k, n = "knnties", int(g["knnties_k"][0]), len(g["knnties_theta"])
    m = O.OracleMap(n, k)
    m.store(g[tag + "_pos"], g[tag + "_power"], g[tag + "_theta"], g[tag + "_phi"])
    m.build(1)
    q = g[tag + "_q_p"]
    fl, fo, r2, idx = m.gather(q, g[tag + "_q_n"], g[tag + "_q_wo"], g[tag + "_q_R"], g[tag + "_q_radius2"], brute=True)
    m.close()
    assert (fo == g[tag + "_found"]).all()
    assert bits_equal(r2, g[tag + "_radius2_out"])
    np.testing.assert_allclose(fl, g[tag + "_flux"], rtol=1e-5)
    pos = g[tag + "_pos"]
    for i in range(len(q)):
        sel = idx[i][idx[i] >= 0]
        d = pos[sel] - q[i]
        d2 = np.sort((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32))
        assert bits_equal(d2, g[tag + "_dist2"][i][: len(d2)]), i


@pytest.mark.parametrize("name", ["box", "sphere", "dragon"])
@pytest.mark.parametrize("faithful", [0, 1])
def test_ray_hits_match_upstream(name, faithful):
    """Scene::intersect through upstream's loader, BVHs, triangle and sphere tests: hit / miss, t, p, n,
    pError bit for bit, and the object hit."""
    g = golden(f"ref_rays_{name}.npz")
    T = f"rays_{name}_"
    sc = Scene.load(asset_scene(f"cornell_{name}.scene"))
    osc = O.OracleScene(sc, 64, 64)
    n = 6000 if (name == "dragon" and faithful) else len(g[T + "hit"])   # the far-first order is slow
    prim, t, p, nrm, perr = osc.trace(g[T + "org"][:n], g[T + "dir"][:n], faithful=faithful)
    osc.close()
    hit = g[T + "hit"][:n] == 1
    assert ((prim >= 0) == hit).all()
    assert bits_equal(t[hit], g[T + "t"][:n][hit])
    for mine, key in ((p, "p"), (nrm, "n"), (perr, "perr")):
        assert bits_equal(mine[hit], g[T + key][:n][hit]), key
    assert (object_of_prim(sc, prim) == g[T + "obj"][:n]).all()


@pytest.mark.parametrize("tag", list(MATERIALS))
@pytest.mark.parametrize("mode", [0, 1])
def test_sample_f_matches_upstream(tag, mode):
    g = golden("ref_materials.npz")
    m = tk_material(tag)
    wo, nn, u = g["mat_wo"], g["mat_n"], g["mat_u"]
    n = len(wo)
    wi, f, pdf = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty(n, np.float32)
    L.orc_sample_f(C.byref(m), mode, n, O.fp(wo), O.fp(nn), O.fp(u), O.fp(wi), O.fp(f), O.fp(pdf))
    t = f"mat_{tag}_{'radiance' if mode == 0 else 'importance'}"
    if tag == "diffuse":
        # the disk mapping calls cosf / sinf, whose last bit belongs to the platform's libm: <= 4 ulp
        # on the direction (99 % of the samples are bit-equal), pdf = cos / pi follows
        assert ulp_distance(wi, g[t + "_wi"]).max() <= 4 or np.abs(wi - g[t + "_wi"]).max() < 3e-7
        assert (bits_equal(wi, g[t + "_wi"], per_row=True)).mean() > 0.97
        np.testing.assert_allclose(pdf, g[t + "_pdf"], rtol=2e-6, atol=1e-7)
    else:
        assert bits_equal(wi, g[t + "_wi"]) and bits_equal(pdf, g[t + "_pdf"])
    assert bits_equal(f, g[t + "_f"])


@pytest.mark.parametrize("W", [512, 784, 1024])
def test_camera_matches_upstream(W):
    """Camera's rasterToCamera from the same cameraToScreen, and generateRay."""
    g = golden("ref_camera.npz")
    sc = Scene.load(asset_scene("cornell_sphere.scene"))
    c2w, r2c = sc.camera_matrices(W, W)
    assert bits_equal(c2w, g["cam_in_c2w"])
    assert bits_equal(r2c, g[f"camera{W}_raster2cam"])
    osc = O.OracleScene(sc, W, W)
    o, d = osc.camera_rays(g[f"camera{W}_raster"])
    osc.close()
    assert bits_equal(o, g[f"camera{W}_org"]) and bits_equal(d, g[f"camera{W}_dir"])


@pytest.mark.parametrize("name", ["box", "sphere", "dragon"])
def test_sample_le_matches_upstream(name):
    """AreaLight::sample_Le: upstream picks the triangle of the light mesh with an unseeded random
    number, so each of its outputs must equal the oracle's for one of the possible picks.  Point,
    normal, radiance and position pdf bit for bit; the direction comes from the disk mapping's
    cosf / sinf (last bit platform dependent; z = sqrt(1 - x^2 - y^2) amplifies it near grazing)."""
    g = golden(f"ref_rays_{name}.npz")
    T = f"rays_{name}_le_"
    sc = Scene.load(asset_scene(f"cornell_{name}.scene"))
    osc = O.OracleScene(sc, 64, 64)
    n = len(g[T + "u1"])
    matched = np.zeros(n, bool)
    for upick in light_pick_candidates(sc, 0):
        out = [np.empty((n, 3), np.float32) for _ in range(4)] + [np.empty(n, np.float32) for _ in range(2)]
        up = np.full(n, upick, np.float32)
        L.orc_sample_le(osc.h, 0, n, O.fp(g[T + "u1"]), O.fp(g[T + "u2"]), O.fp(up), *[O.fp(a) for a in out])
        org, d, nl, le, pp, pd = out
        same = (bits_equal(org, g[T + "org"], per_row=True) & bits_equal(nl, g[T + "n"], per_row=True) &
                bits_equal(le, g[T + "L"], per_row=True) & (pp == g[T + "pdf"][:, 0]) &
                (np.abs(d - g[T + "dir"]).max(1) < 1e-6) & (np.abs(pd - g[T + "pdf"][:, 1]) <= 2e-5 * np.abs(pd) + 1e-7))
        matched |= same
    osc.close()
    assert matched.all(), int((~matched).sum())


@pytest.mark.parametrize("name", ["box", "sphere", "dragon"])
def test_one_light_sample_matches_upstream(name):
    """Scene::uniformSampleOneLight at the hits of the first rays of the ray fixture, with the four
    numbers it takes from its sampler given: light pick, AreaLight::sample_Li, the shadow ray of
    spawnRayTo, intersectP, Material::f.  The returned spectrum bit for bit (no libm call on the way)."""
    g, r = golden("ref_nee.npz"), golden(f"ref_rays_{name}.npz")
    u, want = g[f"nee_{name}_u"], g[f"nee_{name}_ld"]
    n = len(u)
    sc = Scene.load(asset_scene(f"cornell_{name}.scene"))
    osc = O.OracleScene(sc, 64, 64)
    for faithful in (0, 1):
        m = n if not (name == "dragon" and faithful) else 2000
        prim, ld = osc.sample_one_light(r[f"rays_{name}_org"][:m], r[f"rays_{name}_dir"][:m], u[:m], faithful=faithful)
        assert ((prim >= 0) == (r[f"rays_{name}_hit"][:m] == 1)).all()
        assert bits_equal(ld, want[:m])
    osc.close()
    lit = (want != 0).any(axis=1)
    assert 0.2 < lit.mean() < 0.6      # neither all shadowed nor all lit: both outcomes are in the fixture


def test_trace_tile_matches_upstream():
    """SppmTracer::traceTile on given photon maps with an unjittered sampler, two iterations: camera
    paths, next-event estimation, both radiance estimates and the SPPMPixel update.  Ld, photon
    counts and radii equal bit for bit; fluxes to 1e-5 (the order of the sum over the k photons)."""
    g = golden("ref_trace_tile.npz")
    W, H, kg, kc, rec, ng, nc = [int(v) for v in g["tile_meta"]]
    sc = Scene.load(asset_scene("cornell_sphere.scene"))
    osc = O.OracleScene(sc, W, H)
    for split in (0, 1):
        orr = O.OracleSppm(osc, sc.sppm_param(globalSize=ng, causticSize=nc, numRecurse=rec, globalSample=kg, causticSample=kc),
                           sc.tracer.max_depth, sc.tracer.russian_roulette, 0x5EED, threads=4, split_mode=split)
        orr.set_jitter(False)
        orr.set_map_photons(0, g["tile_g_pos"], g["tile_g_power"], g["tile_g_theta"], g["tile_g_phi"], 0)
        orr.set_map_photons(1, g["tile_c_pos"], g["tile_c_power"], g["tile_c_theta"], g["tile_c_phi"], 0)
        orr.build_maps()
        assert orr.counters()[2] == int(g["tile_num_recurse"][0])
        for it in range(2):
            orr.camera_pass(it)
            assert_pixels_match(orr.pixels(), g[f"tile_pixels_iter{it}"])
        orr.close()
    osc.close()


def test_spawn_ray_matches_upstream():
    """Intersection::spawnRay / spawnRayTo: restated in numpy from the oracle's next_up / next_down
    rule, compared with upstream's origins."""
    g = golden("ref_rays_sphere.npz")
    a, out = g["rays_sphere_spawn_in"], g["rays_sphere_spawn_out"]
    p, pe, n, w, tg = (a[:, i] for i in range(5))

    def offset(p, pe, n, w):
        f = np.float32
        d = (np.abs(n) * pe)
        d = (d[:, 0] + d[:, 1]).astype(f) + d[:, 2]
        off = (d[:, None] * n).astype(f)
        dot = (w * n)
        dot = (dot[:, 0] + dot[:, 1]).astype(f) + dot[:, 2]
        off = np.where((dot < 0)[:, None], -off, off)
        po = (p + off).astype(f)
        up = np.nextafter(po, f(np.inf))
        dn = np.nextafter(po, f(-np.inf))
        return np.where(off > 0, up, np.where(off < 0, dn, po)).astype(f)

    o1 = offset(p, pe, n, w)
    o2 = offset(p, pe, n, (tg - p).astype(np.float32))
    assert bits_equal(o1, out[:, 0]) and bits_equal(o2, out[:, 1])
    assert bits_equal((tg - o2).astype(np.float32), out[:, 2])
