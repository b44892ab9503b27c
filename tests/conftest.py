"""pytest configuration: the `gpu` marker and shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def sphere_scene():
    from pathtracer_amd.scene import Scene, asset_scene
    return Scene.load(asset_scene("cornell_sphere.scene"))


@pytest.fixture(scope="session")
def box_scene():
    from pathtracer_amd.scene import Scene, asset_scene
    return Scene.load(asset_scene("cornell_box.scene"))


@pytest.fixture(scope="session")
def dragon_scene():
    from pathtracer_amd.scene import Scene, asset_scene
    return Scene.load(asset_scene("cornell_dragon.scene"))


@pytest.fixture(scope="session")
def gpu_ctx():
    """One device context shared by the GPU tests; fails loudly if the HIP library is missing."""
    from pathtracer_amd.tracer import Context
    ctx = Context(0)
    yield ctx
    ctx.close()


def make_photons(rng, n, surface=True):
    """Synthetic photon cloud: a floor, a wall and some volume scatter, like a Cornell box."""
    pos = (rng.random((n, 3)).astype(np.float32) * 4 - 2)
    if surface:
        u = rng.random(n)
        pos[u < 0.45, 1] = -2.0
        pos[(u >= 0.45) & (u < 0.8), 0] = 2.0
    power = rng.random((n, 3)).astype(np.float32)
    theta = rng.integers(0, 256, n).astype(np.uint8)
    phi = rng.integers(0, 256, n).astype(np.uint8)
    return pos, power, theta, phi


def make_queries(rng, nq, radius2=(0.08, 0.8, 0.005)):
    p = (rng.random((nq, 3)).astype(np.float32) * 4 - 2)
    p[: nq // 2, 1] = -2.0
    n = np.tile(np.array([[0, 1, 0]], np.float32), (nq, 1))
    n[nq // 2:] = np.array([-1, 0, 0], np.float32)
    wo = rng.normal(size=(nq, 3)).astype(np.float32)
    R = rng.random((nq, 3)).astype(np.float32)
    r2 = rng.choice(np.array(radius2, np.float32), nq).astype(np.float32)
    return p, n, wo, R, r2
