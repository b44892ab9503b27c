"""Helpers shared by the tests that read tests/golden/ref_*.npz (outputs of the
upstream renderer's own functions, see oracle/make_ref_golden.py)."""
import numpy as np

from pathtracer_amd import _abi

# the materials oracle/ref_probe.cpp builds: type flag, R, T, etaA, etaB
MATERIALS = {
    "diffuse": (1, (0.725, 0.71, 0.68), (0, 0, 0), 1.0, 1.0),
    "mirror": (2, (1.0, 0.9, 0.8), (0, 0, 0), 1.0, 1.0),
    "glass": (4, (1.0, 1.0, 1.0), (0.9, 1.0, 0.95), 1.0, 1.75496),
}
# the draws of the sampler fixture: get2D, get2D, then (get1D, get2D, get1D, get2D) twice
TILE_DRAWS = [2, 2] + [1, 2, 1, 2] * 2


def tk_material(tag):
    ty, R, T, ea, eb = MATERIALS[tag]
    m = _abi.TkMaterial()
    m.type = ty
    for i in range(3):
        m.R[i], m.T[i] = R[i], T[i]
    m.etaA, m.etaB = ea, eb
    return m


def bits_equal(a, b, per_row=False):
    """Bit-for-bit equality of float arrays (so that -0 != +0 and NaN == NaN of the same payload)."""
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    assert a.shape == b.shape, (a.shape, b.shape)
    eq = a == b
    if per_row:
        return eq.reshape(eq.shape[0], -1).all(1)
    return bool(eq.all())


def ulp_distance(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def sorted_idx(idx):
    """Ascending photon indices per query, -1 padding last."""
    big = np.where(idx < 0, np.iinfo(np.int32).max, idx)
    big = np.sort(big, axis=1)
    return np.where(big == np.iinfo(np.int32).max, -1, big).astype(np.int32)


def object_of_prim(scene, prim):
    """Index of the scene object a primitive id belongs to (-1 for a miss)."""
    out = np.full(len(prim), -1, np.int32)
    for j in range(scene.n_objects):
        ob = scene.objects[j]
        if ob.kind == 0:
            sel = (prim >= ob.first) & (prim < ob.first + ob.count)
        else:
            sel = (prim >= scene.n_triangles + ob.first) & (prim < scene.n_triangles + ob.first + ob.count)
        out[sel] = j
    return out


def light_pick_candidates(scene, light):
    """One triangle-pick random number per triangle of the light's mesh (mid-points of equal
    shares: the area distribution of the Cornell light's two equal triangles is uniform)."""
    ob = scene.objects[scene.lights[light].object]
    return [(i + 0.5) / ob.count for i in range(ob.count)]


def assert_pixels_match(px, ref):
    """SPPMPixel arrays (n, 13): Ld, photon counts and radii bit for bit, the two fluxes to 1e-5."""
    assert bits_equal(px[:, 0:3], ref[:, 0:3]), "Ld"
    assert bits_equal(px[:, 9:13], ref[:, 9:13]), "photon counts / radii"
    np.testing.assert_allclose(px[:, 3:9], ref[:, 3:9], rtol=1e-5, atol=1e-12)


def assert_paths_match(g, prefix, pos, power, theta, phi, shot):
    """A photon pass against upstream's light-path loops (ref_photon_paths.npz): the pass stopped at the path that
    filled the map upstream, every direction byte pair is equal, positions and powers are equal up to the last bit
    of the float trigonometry (most of them bit for bit)."""
    count = g[prefix + "_count"]
    assert shot == int(np.nonzero(count)[0][-1]) + 1, "the path that stored the last photon"
    assert len(pos) == int(count.sum())
    rp, rw, rd = g[prefix + "_pos"], g[prefix + "_power"], g[prefix + "_dir"]
    assert (np.stack([theta, phi], 1) == rd).all(), "direction bytes"
    assert bits_equal(pos, rp, per_row=True).mean() > 0.95 and bits_equal(power, rw, per_row=True).mean() > 0.95
    np.testing.assert_allclose(pos, rp, rtol=0, atol=5e-5)
    np.testing.assert_allclose(power, rw, rtol=1e-5, atol=1e-12)


def float_image(px, counters, W, H):
    """The colour of every pixel before clamp and gamma (src/tracer/TkSppmTracer.cpp:522-524) from an SPPMPixel array
    (n, 13) and (shootGlobal, shootCaustic, numRecurse); the caustic term only when caustic photons were shot."""
    sg, sc, nr = counters
    px = px.astype(np.float64)
    img = px[:, 0:3] / nr + px[:, 6:9] / (np.pi * px[:, 12:13] * sg)
    if sc > 0:
        img = img + px[:, 3:6] / (np.pi * px[:, 11:12] * sc)
    return img.reshape(H, W, 3)


def hsppm_state_close(px, g, prefix, exact):
    """Pixel state of the hash-grid variant (Ld 0:3, Tau 3:6, N 9, radius 11) against upstream's after the same
    iteration.  exact: bit for bit (Tau 1e-5); otherwise all but a per cent of the pixels."""
    ld, tau, N, rad = g[prefix + "Ld"], g[prefix + "Tau"], g[prefix + "N"], g[prefix + "radius"]
    same_n = px[:, 9] == N
    same_r = bits_equal(px[:, 11], rad, per_row=True)
    same_ld = bits_equal(px[:, 0:3], ld, per_row=True)
    close_tau = np.isclose(px[:, 3:6], tau, rtol=1e-5, atol=1e-12).all(1)
    if exact:
        assert same_n.all() and same_r.all() and same_ld.all() and close_tau.all()
    else:
        for m in (same_n, same_r, same_ld, close_tau):
            assert m.mean() >= 0.99, m.mean()
