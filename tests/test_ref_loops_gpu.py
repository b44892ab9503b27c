"""The HIP path against the upstream renderer's own loops and whole renders (tests/golden/ref_photon_paths.npz,
ref_pt_li.npz, ref_hsppm.npz, ref_image_*.npz; see tests/test_ref_loops_cpu.py for how upstream was made to draw
the random numbers the kernels draw).  Every call goes through the C ABI of libtkhip.so."""
import numpy as np
import pytest

from pathtracer_amd import _abi
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import HashGridSppm, PathTracer, SppmTracer
from tests.conftest import golden
from tests.ref_fixtures import assert_paths_match, float_image, hsppm_state_close

pytestmark = pytest.mark.gpu
SEED = 0x5EED


@pytest.mark.parametrize("file,tag", [("cornell_box.scene", "box"), ("cornell_sphere.scene", "sphere"),
                                      ("cornell_dragon.scene", "dragon")])
def test_photon_pass_stores_what_upstream_loops_store(file, tag):
    """k_photon_first / k_photon_paths + the ordered cut against SppmTracer::tracePhoton / traceCausticPhoton run
    path by path (src/tracer/TkSppmTracer.cpp:104-217): same last path, same direction bytes, same photons."""
    g = golden("ref_photon_paths.npz")
    scene = Scene.load(asset_scene(file))
    ng, nc = int(g[f"paths_{tag}_global_count"].sum()), int(g[f"paths_{tag}_caustic_count"].sum())
    tr = SppmTracer.from_scene(scene, globalSize=ng, causticSize=nc, numRecurse=4, device=0, seed=SEED)
    tr.setScene(scene)
    tr.setCamera(32, 32)
    tr.startRaytracing()
    st = _abi.TkIterStats()
    tr.photon_pass(0, st)
    for which, kind, n, shot in ((0, "global", ng, st.global_paths), (1, "caustic", nc, st.caustic_paths)):
        if n == 0:
            continue
        pos, power, th, ph = tr.photons(which)
        assert_paths_match(g, f"paths_{tag}_{kind}", pos, power, th, ph, shot)
    tr.close()


@pytest.mark.parametrize("file,tag", [("cornell_box.scene", "box"), ("cornell_sphere.scene", "sphere")])
def test_path_tracer_sample_matches_upstream_li(file, tag):
    """One unjittered sample of every pixel through the camera-path kernels in path-tracer mode against
    PathTracer::Li (src/tracer/TkPathTracer.cpp:149-204): the film sums are upstream's radiance (1e-5: the kernels
    add a path's contributions hit by hit)."""
    g = golden("ref_pt_li.npz")
    W, spp, depth, xs, ys = (int(v) for v in g[f"li_{tag}_meta"])
    scene = Scene.load(asset_scene(file))
    pt = PathTracer(scene, spp, depth, 0.8, W, W, seed=SEED)
    pt.set_jitter(False)
    pt.render_samples(1)
    contrib, wsum = pt.film()
    ref = g[f"li_{tag}_L"]
    assert (wsum == 1).all()
    assert (np.abs(contrib - ref) <= 1e-5 * np.abs(ref) + 1e-6 * np.abs(ref).max()).all(), float(np.abs(contrib - ref).max())
    pt.close()


def test_hash_grid_sppm_matches_upstream_passes():
    """tkhip_hsppm_* against tk::SPPM's ray pass, grid rebuild, photon pass and renew pass (src/TkSPPM.cpp:45-272)."""
    g = golden("ref_hsppm.npz")
    W, iters, photons, per, numit, spp = (int(v) for v in g["hsppm_sphere_meta"])
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    prm = scene.sppm_param(globalSize=photons, causticSize=0, numRecurse=4, radius2=float(g["hsppm_sphere_radius0"][0]))
    hg = HashGridSppm(scene, prm, spp, 20, 0.8, W, W, seed=SEED)
    hg.set_jitter(False)
    for it in range(iters):
        hg.iteration(it)
        hsppm_state_close(hg.pixels(), g, f"hsppm_sphere_it{it}_", exact=it == 0)
    hg.close()


@pytest.mark.parametrize("tag", ["box", "sphere"])
def test_image_rmse_against_upstream_renders(tag):
    """The final image against SppmTracer::render itself at the same size, photon counts and iteration count:
    the fixture is the mean and run-to-run variance of 8 upstream renders (unseeded, threaded: no two agree).
    Stated bound: a GPU render is within 1.22 sigma * sqrt(1 + 1/8) RMSE of that mean, the mean of 8 GPU renders
    (8 seeds) within 1.22 sigma * sqrt(1/8 + 1/8), sigma^2 = mean per-pixel variance of the clamped colour."""
    g = golden(f"ref_image_{tag}.npz")
    W, H, recurse, gsize, csize, runs, threads = (int(v) for v in g["meta"])
    scene = Scene.load(asset_scene(str(g["scene"])))
    mean, var = g["mean_clipped"].astype(np.float64), g["var_clipped"].astype(np.float64)
    sigma2 = var.mean()
    imgs = []
    for seed in range(SEED, SEED + 8):
        tr = SppmTracer.from_scene(scene, globalSize=gsize, causticSize=csize, numRecurse=recurse, device=0, seed=seed)
        tr.setScene(scene)
        tr.setCamera(W, H)
        tr.startRaytracing()
        for it in range(recurse):
            tr.iteration(it, with_stats=False)
        imgs.append(np.clip(float_image(tr.pixels(), tr.counters(), W, H), 0, 1))
        tr.close()
    for im in imgs:
        assert ((im - mean) ** 2).mean() <= 1.5 * sigma2 * (1 + 1 / runs)
    avg = np.stack(imgs).mean(0)
    mse = ((avg - mean) ** 2).mean()
    assert mse <= 1.5 * sigma2 * (1 / len(imgs) + 1 / runs), (mse, sigma2)
    assert abs(avg.mean() - mean.mean()) <= 0.01
    # and the variance from run to run is upstream's: the renders are draws from the same distribution
    gv = np.stack(imgs).var(0, ddof=1).mean()
    assert 0.6 * sigma2 <= gv <= 1.6 * sigma2, (gv, sigma2)
