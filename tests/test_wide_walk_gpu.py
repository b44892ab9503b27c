"""The wide walk (eight lanes to a ray in the tree of eight-children nodes), which the path kernels use for the last
rays of a wave.  Its hits must be those of the one-lane walk: the same boxes, the same primitive tests.  Checked on
the reference's own ray fixtures with every ray sent that way, and by running the parity suites of the passes (light
paths, camera paths, whole iterations, the path tracer, the hash-grid variant) in a child process whose waves take the
wide walk for every ray they begin once the pass's queue is empty, however many they hold (the threshold is read from
the environment when a context is made; the test scenes' queues are empty almost at once)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd.scene import Scene, asset_scene
from tests.conftest import golden
from tests.ref_fixtures import bits_equal, object_of_prim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["box", "sphere", "dragon"])
def test_wide_walk_hits_match_upstream(gpu_ctx, name):
    g = golden(f"ref_rays_{name}.npz")
    T = f"rays_{name}_"
    sc = Scene.load(asset_scene(f"cornell_{name}.scene"))
    gpu_ctx.set_scene(sc)
    try:
        gpu_ctx.set_wide_walk(0)
        one = gpu_ctx.trace_rays(g[T + "org"], g[T + "dir"])
        gpu_ctx.set_wide_walk(64)
        prim, t, p, nrm, perr = gpu_ctx.trace_rays(g[T + "org"], g[T + "dir"])
        # rays cut short by a t_max of their own (shadow rays): every third ray ends before its hit
        tm = np.where(np.arange(len(t)) % 3 == 0, np.maximum(t * 0.5, 1e-3), 3.0e38).astype(np.float32)
        cut_wide = gpu_ctx.trace_rays(g[T + "org"], g[T + "dir"], tm)
        gpu_ctx.set_wide_walk(0)
        cut_one = gpu_ctx.trace_rays(g[T + "org"], g[T + "dir"], tm)
    finally:
        gpu_ctx.set_wide_walk(12)
    hit = g[T + "hit"] == 1
    assert ((prim >= 0) == hit).all()
    assert bits_equal(t[hit], g[T + "t"][hit])
    for mine, key in ((p, "p"), (nrm, "n"), (perr, "perr")):
        assert bits_equal(mine[hit], g[T + key][hit]), key
    assert (object_of_prim(sc, prim) == g[T + "obj"]).all()
    for a, b in zip(one, (prim, t, p, nrm, perr)):
        assert a.tobytes() == b.tobytes()
    for a, b in zip(cut_one, cut_wide):
        assert a.tobytes() == b.tobytes()
    assert (cut_wide[0] < 0).sum() > (prim < 0).sum()      # the cut did end rays


@pytest.mark.parametrize("rays", ["64", "0"])
def test_pass_parity_suites_with_other_thresholds(rays):
    """64: every ray begun after the queue is empty takes the wide walk; 0: none does."""
    env = dict(os.environ, TKHIP_WIDE_RAYS=rays)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
           "tests/test_parity_gpu.py", "tests/test_ref_pinned_gpu.py", "tests/test_pathtracer_gpu.py", "tests/test_hashgrid_gpu.py",
           "-k", "(photon or iteration or trace_tile or camera or image or pt or path or hash or grid or sharding or rank) and not full_size"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
