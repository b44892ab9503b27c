"""Progressive display: the device-resident frame buffer after iteration i against the oracle's
resolve after iteration i (SppmTracer::rendering, src/tracer/TkSppmTracer.cpp:43-102; saving
mid-render, src/TkRenderer.cpp:220-222 / src/core/TkRayTracer.cpp:124-140)."""
import ctypes as C
import os

import numpy as np
import pytest

from pathtracer_amd import _abi
from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import SppmTracer
from tests.test_parity_gpu import _tracers

pytestmark = pytest.mark.gpu


def _rgb_rows_up(frame):
    """RGBA frame buffer (raster rows) -> the RGB image as the PPM has it (rows flipped)."""
    return frame[::-1, :, :3]


@pytest.mark.parametrize("name,cs", [("cornell_sphere.scene", 1500), ("cornell_box.scene", 0)])
def test_preview_after_each_iteration_matches_oracle(name, cs):
    """Same iterations on both sides; after each one the frame buffer the device refreshed on its own
    (no call but the iteration) equals the oracle's tone-mapped image to 1 LSB (powf's last bit)."""
    scene = Scene.load(asset_scene(name))
    W = 64
    tr, osc, orr = _tracers(scene, W, 8000, cs, 4)
    tr.set_preview(1)
    frame, shown = tr.preview()
    assert shown == -1 and not frame.any()
    for it in range(3):
        tr.iteration()
        orr.iteration(it)
        frame, shown = tr.preview()
        assert shown == it
        assert (frame[:, :, 3] == 255).all()
        oimg = orr.resolve()
        diff = np.abs(_rgb_rows_up(frame).astype(int) - oimg.astype(int))
        assert diff.max() <= 1, (it, int(diff.max()))
        assert (diff == 0).mean() > 0.99
        # the host-side resolve of the same state is the same picture
        assert np.array_equal(_rgb_rows_up(frame), tr.resolve())
    tr.close(); orr.close(); osc.close()


def test_preview_interval_and_refresh_on_demand():
    scene = Scene.load(asset_scene("cornell_box.scene"))
    tr, osc, orr = _tracers(scene, 32, 3000, 0, 6)
    tr.set_preview(2)
    shown = []
    for _ in range(5):
        tr.iteration()
        shown.append(tr.preview()[1])
    assert shown == [-1, 1, 1, 3, 3]
    stale, _ = tr.preview()
    fresh, it = tr.preview(refresh=True)        # the picture of the state as it is now
    assert np.array_equal(_rgb_rows_up(fresh), tr.resolve())
    assert not np.array_equal(stale, fresh)
    tr.set_preview(0)
    tr.iteration()
    assert np.array_equal(tr.preview()[0], fresh)   # switched off: nothing refreshes it
    tr.close(); orr.close(); osc.close()


def test_windowed_flow_saves_mid_render(tmp_path):
    """The C++ tracer class driven as the viewer drives it: updateScreen() per frame, 's' after two
    iterations.  The mid-render PPM and the frame buffer are the oracle's image after two iterations,
    the final PPM the headless render's."""
    text = open(asset_scene("cornell_sphere.scene")).read()
    text = text.replace('<gsize i="200000"/>', '<gsize i="6000"/>').replace('<csize i="50000"/>', '<csize i="800"/>')
    text = text.replace('<recurse i="16"/>', '<recurse i="4"/>')
    text = text.replace("../models/", os.path.join(os.path.dirname(asset_scene("x")), "../models/"))
    sf = tmp_path / "small.scene"
    sf.write_text(text)
    lib = _abi.load_tkhost()
    scene = Scene.load(str(sf))
    W = 56
    mid, out, ref = tmp_path / "mid.ppm", tmp_path / "out.ppm", tmp_path / "ref.ppm"
    frame = np.zeros((W, W), np.uint32)
    rc = lib.tkhost_render_windowed(scene._h, 0, W, W, os.fsencode(str(out)), 0x5EED, 2, os.fsencode(str(mid)),
                                    frame.ctypes.data)
    assert rc == 0, lib.tkhost_last_error()
    assert lib.tkhost_render(scene._h, 0, W, W, os.fsencode(str(ref)), 0x5EED, None) == 0
    assert out.read_bytes() == ref.read_bytes()
    head = b"P6\n%d %d\n255\n" % (W, W)
    data = mid.read_bytes()
    assert data.startswith(head)
    mid_img = np.frombuffer(data[len(head):], np.uint8).reshape(W, W, 3)
    assert np.array_equal(mid_img, _rgb_rows_up(frame.view(np.uint8).reshape(W, W, 4)))

    tr, osc, orr = _tracers(scene, W, 6000, 800, 4)
    for it in range(2):
        orr.iteration(it)
    diff = np.abs(mid_img.astype(int) - orr.resolve().astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.99
    tr.close(); orr.close(); osc.close()
