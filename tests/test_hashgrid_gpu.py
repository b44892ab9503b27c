"""The hash-grid SPPM variant (tk::SPPM, src/TkSPPM.cpp) on the GPU against its restatement in the oracle.
The class is not part of the reference's build, but it compiles with it: tests/test_ref_loops_cpu.py and
tests/test_ref_loops_gpu.py hold the oracle and the kernels to its own passes (tests/golden/ref_hsppm.npz).
Bars: Ld, N and the radius of every pixel bit for bit (the photon counts M behind them are integers);
Tau to 1e-5 relative (photons arrive in another order, so the float sums differ in their last bits)."""
import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import HashGridSppm

pytestmark = pytest.mark.gpu


def _pair(scene, W, photons, radius, iters, spp=1, seed=0x5EED):
    prm = scene.sppm_param(globalSize=photons, causticSize=0, numRecurse=iters, radius2=radius)
    hg = HashGridSppm(scene, prm, spp, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W, seed=seed)
    osc = O.OracleScene(scene, W, W)
    orr = O.OracleSppm(osc, prm, scene.tracer.max_depth, scene.tracer.russian_roulette, seed, threads=8, split_mode=1, faithful_bvh=0)
    return hg, osc, orr


def _assert_state_equal(g, o):
    gu, ou = g.view(np.uint32), o.view(np.uint32)
    assert (gu[:, 0:3] == ou[:, 0:3]).all(), "Ld"
    assert (gu[:, 9] == ou[:, 9]).all(), "N"
    assert (gu[:, 11] == ou[:, 11]).all(), "radius"
    a, b = g[:, 3:6], o[:, 3:6]
    assert (np.abs(a - b) <= 1e-5 * np.abs(b) + 1e-7 * np.abs(b).max()).all(), float(np.abs(a - b).max())


@pytest.mark.parametrize("name,spp", [("cornell_sphere.scene", 1), ("cornell_box.scene", 2), ("cornell_dragon.scene", 1)])
def test_hashgrid_iterations_match_oracle(name, spp):
    scene = Scene.load(asset_scene(name))
    W, iters = 48, 3
    hg, osc, orr = _pair(scene, W, 20000, 0.05, 4, spp)
    for it in range(iters):
        st = hg.iteration(it)
        ost = orr.hash_iteration(spp, it)
        assert st.gathers_caustic == ost[0] and ost[0] > 0.5 * W * W      # visible points
        assert st.global_paths == 20000
        _assert_state_equal(hg.pixels(), orr.pixels())
    px = hg.pixels()
    assert (px[:, 9] > 0).mean() > 0.5 and (px[:, 11] < 0.05).mean() > 0.5   # photons arrived, radii shrank
    gi, oi = hg.resolve(), orr.hash_resolve()
    d = np.abs(gi.astype(int) - oi.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.99
    assert gi.mean() > 10
    hg.close(); orr.close(); osc.close()


def test_hashgrid_converges_to_the_kd_tree_render():
    """Both SPPM classes estimate the same image: after a few iterations their means agree within a few percent
    (different estimators of the same radiance; not a parity check)."""
    from pathtracer_amd.tracer import SppmTracer
    scene = Scene.load(asset_scene("cornell_box.scene"))
    W = 64
    prm = scene.sppm_param(globalSize=50000, causticSize=0, numRecurse=9, radius2=0.12)
    hg = HashGridSppm(scene, prm, 1, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W)
    for it in range(9):
        hg.iteration(it)
    a = hg.resolve().astype(float)
    hg.close()
    tr = SppmTracer.from_scene(scene, globalSize=50000, causticSize=0, numRecurse=9, device=0)
    tr.setScene(scene); tr.setCamera(W, W)
    b = tr.render().astype(float)
    tr.close()
    assert abs(a.mean() - b.mean()) < 0.15 * b.mean()


def test_hashgrid_state_checks():
    from pathtracer_amd import _abi
    scene = Scene.load(asset_scene("cornell_box.scene"))
    prm = scene.sppm_param(globalSize=1000, causticSize=0, numRecurse=4, radius2=0.05)
    with pytest.raises(_abi.TkError):
        HashGridSppm(scene, prm, 3, 20, 0.8, 16, 16)          # spp not a power of two
    from pathtracer_amd.tracer import SppmTracer
    tr = SppmTracer.from_scene(scene, globalSize=1000, causticSize=0, numRecurse=4, device=0)
    tr.setScene(scene); tr.setCamera(16, 16); tr.startRaytracing()
    assert tr.ctx.lib.tkhip_hsppm_iteration(tr.ctx.h, 0, None) == -2   # not a hash-grid render
    tr.close()


def test_hashgrid_scene_element_drives_the_cpp_class(tmp_path):
    """<hashgrid i="1"/> in the tracer block: the C++ class tk::SPPM renders the image the C ABI calls give."""
    import os
    from pathtracer_amd import _abi
    text = open(asset_scene("cornell_sphere.scene")).read()
    text = text.replace('<gsize i="200000"/>', '<gsize i="20000"/>').replace('<recurse i="16"/>', '<recurse i="4"/><hashgrid i="1"/>')
    text = text.replace("../models/", os.path.join(os.path.dirname(asset_scene("x")), "../models/"))
    assert '<hashgrid i="1"/>' in text
    sf = tmp_path / "hash.scene"
    sf.write_text(text)
    lib = _abi.load_tkhost()
    sc = Scene.load(str(sf))
    assert sc.tracer.type == b"sppm_hash"
    W = 48
    out = tmp_path / "hash.ppm"
    assert lib.tkhost_render(sc._h, 0, W, W, os.fsencode(str(out)), 0x5EED, None) == 0, lib.tkhost_last_error()
    head = b"P6\n%d %d\n255\n" % (W, W)
    img = np.frombuffer(out.read_bytes()[len(head):], np.uint8).reshape(W, W, 3)
    spp = 1
    while spp < sc.tracer.spp:
        spp *= 2
    hg = HashGridSppm(sc, sc.sppm_param(), spp, sc.tracer.max_depth, sc.tracer.russian_roulette, W, W)
    for it in range(4):
        hg.iteration(it)
    assert np.array_equal(img, hg.resolve())
    hg.close()
