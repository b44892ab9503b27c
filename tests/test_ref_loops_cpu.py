"""The CPU oracle against the upstream renderer's own LOOPS (tests/golden/ref_photon_paths.npz, ref_pt_li.npz,
ref_hsppm.npz, ref_image_*.npz; written by oracle/ref_probe2.cpp, see oracle/make_ref_golden.py).

Upstream draws its Russian-roulette numbers and the triangle pick of a light sample from an unseeded generator
(src/TkMath.h:312-319).  For these fixtures that one function was made to return numbers fed by the prober
(the patch is recorded in each fixture, key *_patch), and the numbers fed are the ones the oracle and the HIP
kernels draw at the same places (oracle_py.photon_draws / pt_draws): upstream's tracePhoton(H) then traces the
path the oracle traces for Halton index H, and the stored photons can be compared one by one.  What is left
between the two is the last bit of the float trigonometry (upstream calls libm's float cos / sin, the oracle and
the kernels round the double function to float so that host and GPU agree), which moves a few per cent of the
positions by an ulp or so: counts and direction bytes are equal, positions and powers close.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd.scene import Scene, asset_scene
from tests.conftest import golden
from tests.ref_fixtures import assert_paths_match, bits_equal, float_image, hsppm_state_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED


@pytest.mark.parametrize("file,tag", [("cornell_box.scene", "box"), ("cornell_sphere.scene", "sphere"),
                                      ("cornell_dragon.scene", "dragon")])
def test_light_path_loops_store_what_upstream_stores(file, tag):
    """tracePhoton / traceCausticPhoton as loops (src/tracer/TkSppmTracer.cpp:104-217): store before roulette, the
    depth caps of 20 / 64, 'a non-diffuse hit before the first diffuse one', the reuse of Halton dimensions 5 and 6.
    The oracle's photon pass with map sizes equal to what upstream's first n paths stored must stop at the same
    path and hold the same photons."""
    g = golden("ref_photon_paths.npz")
    scene = Scene.load(asset_scene(file))
    ng, nc = int(g[f"paths_{tag}_global_count"].sum()), int(g[f"paths_{tag}_caustic_count"].sum())
    osc = O.OracleScene(scene, 32, 32)
    r = O.OracleSppm(osc, scene.sppm_param(globalSize=ng, causticSize=nc, numRecurse=4), 20, 0.8, SEED, threads=4)
    r.photon_pass(0)
    for which, kind, n in ((0, "global", ng), (1, "caustic", nc)):
        if n == 0:
            assert which == 1 and tag == "box"          # nothing specular in the box: no caustic photon, upstream neither
            continue
        pos, power, th, ph, shot = r.photons(which, n)
        assert_paths_match(g, f"paths_{tag}_{kind}", pos, power, th, ph, shot)
    r.close()
    osc.close()


def test_draw_tables_are_what_the_fixture_was_made_with():
    g = golden("ref_photon_paths.npz")
    assert "get_random_float" in str(g["paths_patch"])
    # a path consumed 1 + (bounces it survived to) numbers, never more than its table held
    for k in g.files:
        if k.endswith("_draws_used"):
            assert 1 <= g[k].min() and g[k].max() <= 65
    d = O.photon_draws(SEED, 0, 0, 4, 65)
    assert d.shape == (4, 65) and (d >= 0).all() and (d < 1).all()
    assert d[2, 0] == O.lib().orc_rng_u01(SEED, 2, 0, 0) and d[2, 3] == O.lib().orc_rng_u01(SEED, 2, 0 | (2 << 4), 1)


@pytest.mark.parametrize("file,tag", [("cornell_box.scene", "box"), ("cornell_sphere.scene", "sphere")])
def test_path_tracer_li_matches_upstream(file, tag):
    """PathTracer::Li (src/tracer/TkPathTracer.cpp:149-204) for sample 0 of every pixel, sampler without jitter,
    roulette numbers fed: the radiance is upstream's bit for bit."""
    g = golden("ref_pt_li.npz")
    W, spp, depth, xs, ys = (int(v) for v in g[f"li_{tag}_meta"])
    scene = Scene.load(asset_scene(file))
    osc = O.OracleScene(scene, W, W)
    r = O.OracleSppm(osc, scene.sppm_param(globalSize=1, causticSize=0, numRecurse=1), depth, 0.8, SEED, threads=4)
    r.set_jitter(False)
    contrib, wsum = np.zeros((W * W, 3), np.float32), np.zeros(W * W, np.float32)
    r.pt_render(spp, 0, 1, contrib, wsum)
    assert (wsum == 1).all()
    assert bits_equal(contrib, g[f"li_{tag}_L"])
    r.close()
    osc.close()


def test_hash_grid_cells_match_upstream():
    """SpatialHashGrids::worldToGrid and ::hash (src/TkSPPM.cpp:88-112), incl. points outside the grid and the
    negative products the s32 -> size_t widening turns into large bucket numbers."""
    g = golden("ref_hsppm.npz")
    t = "hsppm_sphere_it0_"
    W = int(g["hsppm_sphere_meta"][0])
    pts = np.ascontiguousarray(g[t + "w2g_points"])
    n = len(pts)
    cell, ins, hv = np.empty((n, 3), np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
    b, res = np.ascontiguousarray(g[t + "grid_bounds"].reshape(6)), np.ascontiguousarray(g[t + "grid_res"])
    O.lib().orc_hg_cells(O.fp(b), O.ip(res), W * W, n, O.fp(pts), O.ip(cell), O.ip(ins), O.ip(hv))
    assert (cell == g[t + "w2g_cell"]).all() and (ins == g[t + "w2g_inside"]).all() and (hv == g[t + "w2g_hash"]).all()
    assert 0 < ins.mean() < 1


def test_hash_grid_sppm_passes_match_upstream():
    """tk::SPPM pass by pass (src/TkSPPM.cpp:136-272): traceTile, SpatialHashGrids::rebuild, tracePhoton with fed
    draws, updatePixel, two iterations.  Iteration 0 is upstream's state bit for bit (Tau to 1e-6); in iteration 1
    a photon whose position differs in the last bit may fall on the other side of a pixel's radius."""
    g = golden("ref_hsppm.npz")
    W, iters, photons, per, numit, spp = (int(v) for v in g["hsppm_sphere_meta"])
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    osc = O.OracleScene(scene, W, W)
    prm = scene.sppm_param(globalSize=photons, causticSize=0, numRecurse=4, radius2=float(g["hsppm_sphere_radius0"][0]))
    r = O.OracleSppm(osc, prm, 20, 0.8, SEED, threads=4)
    r.set_jitter(False)
    for it in range(iters):
        r.hash_iteration(spp, it)
        hsppm_state_close(r.pixels(), g, f"hsppm_sphere_it{it}_", exact=it == 0)
    r.close()
    osc.close()


@pytest.mark.parametrize("tag", ["box", "sphere"])
def test_image_rmse_against_upstream_renders(tag):
    """North star's third criterion: the final image within a stated RMSE of upstream's at a matched iteration
    count.  The fixture holds the mean and the per-pixel variance of 8 runs of SppmTracer::render itself
    (its own threads, its own unseeded random numbers: no two runs agree).  A render of the oracle with another
    random sequence is one more draw from (what should be) the same distribution: its squared distance to the
    8-run mean is expected to be var * (1 + 1/8), and the mean of R such renders lies var * (1/R + 1/8) away."""
    g = golden(f"ref_image_{tag}.npz")
    W, H, recurse, gsize, csize, runs, threads = (int(v) for v in g["meta"])
    scene = Scene.load(asset_scene(str(g["scene"])))
    osc = O.OracleScene(scene, W, H)
    prm = scene.sppm_param(globalSize=gsize, causticSize=csize, numRecurse=recurse)
    imgs = []
    for seed in (SEED, 0xBEEF):
        r = O.OracleSppm(osc, prm, 20, 0.8, seed, threads=8)
        for it in range(recurse):
            r.iteration(it)
        imgs.append(float_image(r.pixels(), r.counters(), W, H))
        r.close()
    osc.close()
    mean, var = g["mean_clipped"].astype(np.float64), g["var_clipped"].astype(np.float64)
    sigma2 = var.mean()
    for im in imgs:
        mse = ((np.clip(im, 0, 1) - mean) ** 2).mean()
        assert mse <= 1.5 * sigma2 * (1 + 1 / runs), (mse, sigma2)          # stated bound: 1.22 sigma, sigma = run-to-run RMS
    both = np.clip(np.stack(imgs), 0, 1).mean(0)
    assert ((both - mean) ** 2).mean() <= 1.5 * sigma2 * (1 / 2 + 1 / runs)
    assert abs(both.mean() - mean.mean()) <= 0.01                            # no overall bias in brightness


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs upstream's sources (build container only)")
def test_fixtures_are_reproduced_by_the_probers():
    """Where upstream's sources are at hand: the probers are rebuilt and every reproducible fixture regenerated;
    each array equals the committed one byte for byte (except the light-emission origins, whose triangle pick
    upstream draws unseeded, and the whole-render images, which no two runs reproduce)."""
    subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")], check=True, capture_output=True, timeout=900)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_ref_golden.py"), "--out", tmp], check=True,
                       capture_output=True, timeout=900, cwd=ROOT)
        names = sorted(os.listdir(tmp))
        assert len(names) >= 12
        for name in names:
            new, old = np.load(os.path.join(tmp, name)), golden(name)
            assert sorted(new.files) == sorted(old.files), name
            for k in new.files:
                if k.endswith("_le_org"):
                    continue
                assert new[k].dtype == old[k].dtype and new[k].tobytes() == old[k].tobytes(), (name, k)
