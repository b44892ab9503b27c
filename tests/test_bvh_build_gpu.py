"""The scene hierarchy built on the GPU against the one built on host threads: the closest hit of a ray
does not depend on the tree, so every hit record must be equal bit for bit, whichever builder ran
(buildBVH, src/BVH.cpp:62-243; the upstream hits are those of tests/golden/ref_rays_*.npz)."""
import numpy as np
import pytest

from pathtracer_amd.scene import Scene, asset_scene
from pathtracer_amd.tracer import Context
from tests.conftest import golden
from tests.ref_fixtures import bits_equal, object_of_prim

pytestmark = pytest.mark.gpu

HOST, DEVICE = 1, 2


def _ctx(builder):
    ctx = Context(0)
    ctx.set_bvh_builder(builder)
    return ctx


@pytest.mark.parametrize("name", ["box", "sphere", "dragon"])
def test_upstream_hits_with_the_tree_built_on_the_gpu(name):
    g = golden(f"ref_rays_{name}.npz")
    T = f"rays_{name}_"
    sc = Scene.load(asset_scene(f"cornell_{name}.scene"))
    ctx = _ctx(DEVICE)
    ctx.set_scene(sc)
    info = ctx.bvh_info()
    assert info["on_device"] and 0 < info["depth"] < 31 and info["nodes"] < sc.n_triangles + sc.n_spheres
    prim, t, p, nrm, perr = ctx.trace_rays(g[T + "org"], g[T + "dir"])
    hit = g[T + "hit"] == 1
    assert ((prim >= 0) == hit).all()
    assert bits_equal(t[hit], g[T + "t"][hit])
    for mine, key in ((p, "p"), (nrm, "n"), (perr, "perr")):
        assert bits_equal(mine[hit], g[T + key][hit]), key
    assert (object_of_prim(sc, prim) == g[T + "obj"]).all()
    ctx.close()


def _random_rays(rng, lo, hi, n):
    org = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.random(n) < 0.3, 0.4 * float(np.max(hi - lo)), 3.0e38).astype(np.float32)
    return org, d, tmax


def _same_hits(sc, org, d, tmax):
    out = {}
    for builder in (HOST, DEVICE):
        ctx = _ctx(builder)
        ctx.set_scene(sc)
        out[builder] = (ctx.trace_rays(org, d, tmax), ctx.bvh_info())
        ctx.close()
    assert not out[HOST][1]["on_device"] and out[DEVICE][1]["on_device"]
    for a, b in zip(out[HOST][0], out[DEVICE][0]):
        assert bits_equal(a, b)
    return out


def test_dragon_hits_do_not_depend_on_the_builder(dragon_scene):
    rng = np.random.default_rng(11)
    org, d, tmax = _random_rays(rng, np.array([-1.0, -1.0, -1.0]), np.array([1.0, 1.0, 1.0]), 200000)
    out = _same_hits(dragon_scene, org, d, tmax)
    assert (out[HOST][0][0] >= 0).mean() > 0.2
    print("dragon build ms: host %.1f, device %.1f; depth %d / %d" % (out[HOST][1]["ms"], out[DEVICE][1]["ms"],
                                                                     out[HOST][1]["depth"], out[DEVICE][1]["depth"]))


def test_height_field_hits_do_not_depend_on_the_builder():
    """A synthetic field of 450 x 450 quads (405 000 triangles, many with equal box centres along rows)."""
    sc = Scene.synthetic(450, seed=3)
    rng = np.random.default_rng(12)
    org, d, tmax = _random_rays(rng, np.array([-1.0, -0.2, -1.0]), np.array([1.0, 1.0, 1.0]), 200000)
    out = _same_hits(sc, org, d, tmax)
    assert (out[HOST][0][0] >= 0).mean() > 0.05


def test_by_default_large_scenes_build_on_the_gpu_and_small_ones_on_the_host(dragon_scene, sphere_scene):
    ctx = Context(0)
    ctx.set_scene(dragon_scene)
    assert ctx.bvh_info()["on_device"]
    ctx.set_scene(sphere_scene)
    assert not ctx.bvh_info()["on_device"]
    ctx.close()


def test_identical_primitives_and_tiny_scenes():
    """Equal Morton codes throughout (every triangle the same) and the smallest scene the GPU builder takes."""
    import ctypes as C
    from pathtracer_amd import _abi
    lib = _abi.load_tkhip()
    for n in (3, 5, 64, 1000):
        tris = (_abi.TkTriangle * n)()
        for i in range(n):
            tris[i].v0[:] = (0.0, 0.0, 1.0)
            tris[i].v1[:] = (1.0, 0.0, 1.0)
            tris[i].v2[:] = (0.0, 1.0, 1.0)
        obj = (_abi.TkObject * 1)()
        obj[0].kind, obj[0].first, obj[0].count, obj[0].material, obj[0].light = 0, 0, n, 0, -1
        mat = (_abi.TkMaterial * 1)()
        mat[0].type = 1   # TK_DIFFUSE
        ctx = _ctx(DEVICE)
        _abi.check(lib, lib.tkhip_set_scene(ctx.h, obj, 1, tris, n, None, 0, mat, 1, None, 0), "tkhip_set_scene")
        assert ctx.bvh_info()["on_device"]
        org = np.array([[0.2, 0.2, 0.0], [2.0, 2.0, 0.0]], np.float32)
        d = np.array([[0.01, 0.02, 1.0], [0.01, 0.02, 1.0]], np.float32)   # no zero component (those miss every box upstream)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        prim, t, *_ = ctx.trace_rays(org, d)
        assert prim[0] >= 0 and abs(t[0] - 1.0 / d[0, 2]) < 1e-5 and prim[1] == -1
        ctx.close()
