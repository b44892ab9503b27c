"""One render over several ranks of ONE process (tkhip_group_*: a context and a host thread per rank, photons
pulled with peer copies), all ranks on this GPU, against the oracle run with the same sharding; and the C++
tracer class / the <gpus> scene element on top of it."""
import os

import numpy as np
import pytest

from oracle import oracle_py as O
from pathtracer_amd import _abi, distributed as D
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import SppmGroup
from tests.test_parity_gpu import _assert_pixels_equal

pytestmark = pytest.mark.gpu


def _oracle_sharded(scene, prm, W, world, iters, seed=0x5EED):
    """The oracle's ranks driven in lockstep with the exchange done by hand; returns the merged SPPMPixel array."""
    osc = O.OracleScene(scene, W, W)
    ors = [O.OracleSppm(osc, prm, scene.tracer.max_depth, scene.tracer.russian_roulette, seed, rank=r, world=world, threads=4)
           for r in range(world)]
    for o in ors:
        o.set_tie_rule(True)
    for it in range(iters):
        for o in ors:
            o.photon_pass(it)
        for which, size in ((0, prm.globalSize), (1, prm.causticSize)):
            op = [o.photons(which, max(size, 1)) for o in ors]
            shot = sum(p[4] for p in op)
            for o in ors:
                o.set_map_photons(which, np.concatenate([p[0] for p in op]), np.concatenate([p[1] for p in op]),
                                  np.concatenate([p[2] for p in op]), np.concatenate([p[3] for p in op]), shot)
        for o in ors:
            o.build_maps()
            o.camera_pass(it)
    px = [o.pixels() for o in ors]
    owner = np.repeat(np.array([D.row_owner(y, world) for y in range(W)]), W)
    merged = px[0].copy()
    for r in range(1, world):
        merged[owner == r] = px[r][owner == r]
    for o in ors:
        o.close()
    osc.close()
    return merged


@pytest.mark.parametrize("world,begin_next", [(2, 0), (2, 1), (3, 1), (2, 2), (3, 2)])
def test_group_iterations_match_the_sharded_oracle(world, begin_next):
    """Ld, photon counts and radii of every pixel bit for bit, fluxes to 1e-5, after three iterations; with
    begin_next 1 the light paths of iteration i+1 run under the gathers of iteration i, with 2 its camera paths
    as well (second set of query buffers, radii read again at the gather), and nothing changes."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W, GS, CS, iters = 64, 8000, 1000, 3
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    grp = SppmGroup(scene, [0] * world, prm, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W)
    for it in range(iters):
        st = grp.iteration(it, begin_next=begin_next if it + 1 < iters else 0, with_stats=True)
        assert sum(s.global_stored for s in st) == GS           # the quotas of the ranks make up the map
    got = grp.pixels()
    want = _oracle_sharded(scene, prm, W, world, iters)
    _assert_pixels_equal(got, want)
    # a merge mid-render changes nothing for the iterations that follow it
    img = grp.resolve(4)
    assert img.shape == (W, W, 3) and img.mean() > 10
    grp.close()


@pytest.mark.parametrize("world", [2, 3])
def test_merge_with_the_next_camera_paths_in_flight(world):
    """begin_next 2 leaves the camera paths of the next iteration adding direct light to the pixels when the call
    returns.  A merge at that moment (a mid-render save) waits for them, moves only the other ranks' rows into
    rank 0's pixels and loses nothing: the iterations that follow end in the sharded oracle's pixel state."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    W, GS, CS = 64, 8000, 1000
    prm = scene.sppm_param(globalSize=GS, causticSize=CS, numRecurse=4)
    grp = SppmGroup(scene, [0] * world, prm, scene.tracer.max_depth, scene.tracer.russian_roulette, W, W)
    grp.iteration(0, begin_next=2)
    mid = grp.pixels()                       # merges with iteration 1's camera paths in flight
    one = _oracle_sharded(scene, prm, W, world, 1)
    assert np.array_equal(mid[:, 9:13], one[:, 9:13])          # photon counts and radii: after one iteration
    np.testing.assert_allclose(mid[:, 3:9], one[:, 3:9], rtol=1e-5, atol=1e-10)
    grp.iteration(1, begin_next=2)
    grp.merge()
    img = grp.resolve(4)
    assert img.mean() > 10
    grp.iteration(2)
    _assert_pixels_equal(grp.pixels(), _oracle_sharded(scene, prm, W, world, 3))
    grp.close()


def test_group_of_one_is_the_single_gpu_render():
    scene = Scene.load(asset_scene("cornell_box.scene"))
    prm = scene.sppm_param(globalSize=5000, causticSize=0, numRecurse=4)
    grp = SppmGroup(scene, [0], prm, scene.tracer.max_depth, scene.tracer.russian_roulette, 48, 48)
    for it in range(2):
        grp.iteration(it)
    from pathtracer_amd.tracer import SppmTracer
    tr = SppmTracer.from_scene(scene, globalSize=5000, causticSize=0, numRecurse=4, device=0)
    tr.setScene(scene); tr.setCamera(48, 48); tr.startRaytracing()
    for _ in range(2):
        tr.iteration()
    assert np.array_equal(grp.pixels(), tr.pixels())
    tr.close(); grp.close()


def test_group_closed_with_the_next_iteration_in_flight():
    """begin_next 2 leaves the light and camera paths of the next iteration running when the call returns;
    closing the render then waits for them before anything is freed, and a new render starts clean."""
    scene = Scene.load(asset_scene("cornell_sphere.scene"))
    prm = scene.sppm_param(globalSize=8000, causticSize=1000, numRecurse=4)
    for _ in range(2):
        grp = SppmGroup(scene, [0, 0], prm, scene.tracer.max_depth, scene.tracer.russian_roulette, 64, 64)
        grp.iteration(0, begin_next=2)
        grp.close()


def test_group_reports_a_failing_rank():
    lib = _abi.load_tkhip()
    scene = Scene.load(asset_scene("cornell_box.scene"))
    prm = scene.sppm_param(globalSize=5000, causticSize=0, numRecurse=4)
    grp = SppmGroup(scene, [0, 0], prm, scene.tracer.max_depth, scene.tracer.russian_roulette, 32, 32)
    grp.iteration(0, begin_next=True)
    with pytest.raises(_abi.TkError) as e:       # the light paths of iteration 1 are in flight, not those of 5
        grp.iteration(5)
    assert "in flight" in str(e.value)
    grp.iteration(1)
    grp.close()


def test_scene_gpus_element_drives_the_cpp_tracer(tmp_path):
    """<gpus i="2"/> in the scene root: the C++ tracer class renders over two ranks (sharing this GPU); the image
    is the one the group produces when driven from Python, and it differs from the single-rank image only as far
    as another split of the same photon budget does."""
    text = open(asset_scene("cornell_sphere.scene")).read()
    text = text.replace('<gsize i="200000"/>', '<gsize i="6000"/>').replace('<csize i="50000"/>', '<csize i="800"/>')
    text = text.replace('<recurse i="16"/>', '<recurse i="4"/>')
    text = text.replace("../models/", os.path.join(os.path.dirname(asset_scene("x")), "../models/"))
    assert "<camera>" in text
    two = tmp_path / "two.scene"
    two.write_text(text.replace("<camera>", '<gpus i="2"/>\n<camera>', 1))
    lib = _abi.load_tkhost()
    sc = Scene.load(str(two))
    assert lib.tkhost_gpu_count(sc._h) == 2
    W = 64
    out = tmp_path / "two.ppm"
    assert lib.tkhost_render(sc._h, 0, W, W, os.fsencode(str(out)), 0x5EED, None) == 0, lib.tkhost_last_error()
    head = b"P6\n%d %d\n255\n" % (W, W)
    img = np.frombuffer(out.read_bytes()[len(head):], np.uint8).reshape(W, W, 3)
    grp = SppmGroup(sc, [0, 0], sc.sppm_param(), sc.tracer.max_depth, sc.tracer.russian_roulette, W, W)
    for it in range(4):
        grp.iteration(it, begin_next=it < 3)
    assert np.array_equal(img, grp.resolve(4))
    grp.close()
    lib.tkhost_set_gpu_count(sc._h, 1)
    one = tmp_path / "one.ppm"
    assert lib.tkhost_render(sc._h, 0, W, W, os.fsencode(str(one)), 0x5EED, None) == 0
    img1 = np.frombuffer(one.read_bytes()[len(head):], np.uint8).reshape(W, W, 3)
    assert not np.array_equal(img, img1) and abs(img.astype(float).mean() - img1.astype(float).mean()) < 4.0
    # the windowed flow on two ranks: the mid-render frame is the merged picture
    frame = np.zeros((W, W), np.uint32)
    lib.tkhost_set_gpu_count(sc._h, 2)
    rc = lib.tkhost_render_windowed(sc._h, 0, W, W, os.fsencode(str(tmp_path / "w.ppm")), 0x5EED, 2, None, frame.ctypes.data)
    assert rc == 0, lib.tkhost_last_error()
    assert (tmp_path / "w.ppm").read_bytes() == out.read_bytes()
    fr = frame.view(np.uint8).reshape(W, W, 4)
    assert (fr[:, :, 3] == 255).all() and fr[:, :, :3].mean() > 10
