"""Regenerates the golden fixtures under tests/golden/ from the CPU oracle.

The reference ships no tests, fixtures or golden vectors and cannot be built
in this image, so these vectors pin the oracle's behaviour (regression) and
give the GPU tests expected outputs that do not need the oracle at run time.
Hand-derived known answers live in the tests themselves.

    python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402
from pathtracer_amd import _abi  # noqa: E402
from pathtracer_amd.scene import Scene, asset_scene  # noqa: E402
from tests.conftest import make_photons, make_queries  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
L = O.lib()


def main():
    L = O.lib()
    rng = np.random.default_rng(20240611)

    # F4: RadicalInverse(dim 0..6, index 0..4095) + a few 40-bit indices
    dim = np.repeat(np.arange(7, dtype=np.int32), 512)
    idx = np.tile(np.arange(512, dtype=np.uint64), 7)
    big = rng.integers(1 << 32, 1 << 40, 64).astype(np.uint64)
    dim = np.concatenate([dim, rng.integers(0, 7, 64).astype(np.int32)])
    idx = np.concatenate([idx, big])
    val = np.array([L.orc_radical_inverse(int(d), int(i)) for d, i in zip(dim, idx)], np.float32)
    np.savez_compressed(os.path.join(OUT, "radical_inverse.npz"), dim=dim, index=idx, value=val)

    # sampler pieces: pixel hash and permutation element
    px = rng.integers(0, 1024, (256, 4)).astype(np.int32)
    px[:, 3] = 0
    hv = np.array([L.orc_hash_pixel(*map(int, r)) for r in px], np.uint64)
    pe_in = np.stack([rng.integers(0, 16, 256), np.full(256, 16), (hv & 0xffffffff)], 1).astype(np.uint64)
    pe = np.array([L.orc_permutation_element(int(a), int(b), int(c)) for a, b, c in pe_in], np.int32)
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), pixel_dim_seed=px, hash=hv, perm_in=pe_in, perm=pe)

    # F5: direction tables
    tabs = [np.empty(256, np.float32) for _ in range(4)]
    L.orc_direction_tables(*[O.fp(t) for t in tabs])
    np.savez_compressed(os.path.join(OUT, "direction_tables.npz"), cos_theta=tabs[0], sin_theta=tabs[1],
                        cos_phi=tabs[2], sin_phi=tabs[3])

    # F1/F2: kNN cases
    for name, n, k, nq in (("knn_surface", 6000, 100, 600), ("knn_small_k50", 1500, 50, 300)):
        pos, power, th, ph = make_photons(rng, n)
        qp, qn, qwo, qR, r2 = make_queries(rng, nq)
        m = O.OracleMap(n, k)
        m.store(pos, power, th, ph)
        m.build(0)
        flux, found, r2o, nidx = m.gather(qp, qn, qwo, qR, r2)
        bflux, bfound, br2o, bidx = m.gather(qp, qn, qwo, qR, r2, brute=True)
        assert (found == bfound).all() and (nidx == bidx).all() and (r2o == br2o).all()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), pos=pos, power=power, theta=th, phi=ph, k=k, qp=qp, qn=qn,
                            qwo=qwo, qR=qR, r2=r2, flux=flux, found=found, r2out=r2o, idx=nidx)
        m.close()
    # ties: duplicated positions and photons exactly at the search radius; expected sets by the index rule
    n, k, nq = 400, 20, 64
    base = (rng.integers(-8, 8, (n // 4, 3)) * 0.25).astype(np.float32)
    pos = np.concatenate([base, base, base, base]).astype(np.float32)
    rng.shuffle(pos)
    power = rng.random((n, 3)).astype(np.float32)
    th = rng.integers(0, 256, n).astype(np.uint8)
    ph = rng.integers(0, 256, n).astype(np.uint8)
    qp = (rng.integers(-8, 8, (nq, 3)) * 0.25).astype(np.float32)
    qn = np.tile(np.array([[0, 0, 1]], np.float32), (nq, 1))
    qwo = np.tile(np.array([[0.3, 0.2, 0.9]], np.float32), (nq, 1))
    qR = np.full((nq, 3), 0.5, np.float32)
    r2 = rng.choice(np.array([0.0625, 0.125, 0.25, 1.0], np.float32), nq).astype(np.float32)
    m = O.OracleMap(n, k)
    m.store(pos, power, th, ph)
    m.build(0)
    flux, found, r2o, nidx = m.gather(qp, qn, qwo, qR, r2, brute=True)
    np.savez_compressed(os.path.join(OUT, "knn_ties.npz"), pos=pos, power=power, theta=th, phi=ph, k=k, qp=qp, qn=qn,
                        qwo=qwo, qR=qR, r2=r2, flux=flux, found=found, r2out=r2o, idx=nidx)
    m.close()

    # F3: SPPMPixel update traces
    nupd = 512
    flux0 = rng.random((nupd, 3)).astype(np.float32) * 10
    photons0 = np.where(rng.random(nupd) < 0.3, 0, rng.random(nupd) * 300).astype(np.float32)
    rad0 = (rng.random(nupd).astype(np.float32) * 0.08 + 1e-3).astype(np.float32)
    lind = rng.random((nupd, 3)).astype(np.float32) * 5
    found = rng.integers(0, 101, nupd).astype(np.int32)
    found[:16] = 0
    r2ret = np.where(found == 100, rad0 * 0.7, rad0).astype(np.float32)
    f1, p1, r1 = flux0.copy(), photons0.copy(), rad0.copy()
    L.orc_pixel_update(nupd, O.fp(f1), O.fp(p1), O.fp(r1), O.fp(lind), O.ip(found), O.fp(r2ret))
    np.savez_compressed(os.path.join(OUT, "pixel_update.npz"), flux=flux0, photons=photons0, radius2=rad0, lindir=lind,
                        found=found, r2ret=r2ret, flux_out=f1, photons_out=p1, radius2_out=r1)

    # F6/F8: camera rays and ray hits on the sphere and box scenes (small); the dragon is checked live
    for sname in ("cornell_sphere", "cornell_box"):
        sc = Scene.load(asset_scene(sname + ".scene"))
        osc = O.OracleScene(sc, 512, 512)
        raster = (rng.random((1500, 2)) * 512).astype(np.float32)
        raster[:4] = [[0, 0], [512, 512], [256, 256], [0.5, 511.5]]
        org, d = osc.camera_rays(raster)
        prim, t, p, nrm, perr = osc.trace(org, d)
        hit = prim >= 0
        d2 = rng.normal(size=(int(hit.sum()), 3)).astype(np.float32)
        d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
        o2 = (p[hit] + 1e-3 * nrm[hit] * np.sign((d2 * nrm[hit]).sum(1, keepdims=True))).astype(np.float32)
        prim2, t2, p2, n2, e2 = osc.trace(o2, d2)
        np.savez_compressed(os.path.join(OUT, f"rays_{sname}.npz"), raster=raster, org=org, dir=d, prim=prim, t=t, p=p,
                            nrm=nrm, perr=perr, org2=o2, dir2=d2, prim2=prim2, t2=t2, p2=p2, nrm2=n2, perr2=e2)
        osc.close()

    materials_and_lights()


def _unit(v):
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def sample_f(material, mode, wo, nrm, u):
    n = len(wo)
    wi = np.zeros((n, 3), np.float32); f = np.zeros((n, 3), np.float32); pdf = np.zeros(n, np.float32)
    L.orc_sample_f(C.byref(material), mode, n, O.fp(wo), O.fp(nrm), O.fp(u), O.fp(wi), O.fp(f), O.fp(pdf))
    return wi, f, pdf


def sample_le(osc, light, u1, u2, upick):
    n = len(upick)
    out = [np.zeros((n, 3), np.float32) for _ in range(4)] + [np.zeros(n, np.float32) for _ in range(2)]
    L.orc_sample_le(osc.h, light, n, O.fp(u1), O.fp(u2), O.fp(upick), *[O.fp(o) for o in out])
    return out


def materials_and_lights():
    """F7 / F9: Material::sample_f of the three materials in both transport modes, and AreaLight::sample_Le of
    the Cornell light.  Own random stream, so that the fixtures above do not move."""
    rng = np.random.default_rng(20261111)
    sc = Scene.load(asset_scene("cornell_sphere.scene"))
    n = 600
    nrm = _unit(rng.normal(size=(n, 3)))
    wo = _unit(rng.normal(size=(n, 3)))
    wo[::7] = _unit(nrm[::7] + 0.05 * rng.normal(size=nrm[::7].shape))     # near-normal incidence
    wo[3::11] = _unit(np.cross(nrm[3::11], wo[3::11]) + 0.01 * nrm[3::11])  # grazing
    u = rng.random((n, 2)).astype(np.float32)
    u[:3] = [[0, 0], [0.999999, 0.999999], [0.5, 0.5]]
    out = {"wo": wo, "nrm": nrm, "u": u}
    kinds = {_abi.TK_DIFFUSE: "diffuse", _abi.TK_MIRROR: "mirror", _abi.TK_GLASS: "glass"}
    seen = set()
    for i in range(sc.n_materials):
        m = sc.materials[i]
        if m.type in seen:
            continue
        seen.add(m.type)
        name = kinds[m.type]
        out[name + "_material"] = np.frombuffer(bytes(m), np.uint8).copy()
        for mode in (0, 1):
            wi, f, pdf = sample_f(m, mode, wo, nrm, u)
            out[f"{name}_wi_{mode}"] = wi; out[f"{name}_f_{mode}"] = f; out[f"{name}_pdf_{mode}"] = pdf
    assert seen == set(kinds)
    np.savez_compressed(os.path.join(OUT, "sample_f.npz"), **out)
    osc = O.OracleScene(sc, 64, 64)
    m = 800
    u1 = rng.random((m, 2)).astype(np.float32); u2 = rng.random((m, 2)).astype(np.float32)
    upick = rng.random(m).astype(np.float32)
    u1[:2] = [[0, 0], [0.999999, 0.999999]]; upick[:2] = [0, 0.999999]
    org, d, nl, le, pp, pd = sample_le(osc, 0, u1, u2, upick)
    np.savez_compressed(os.path.join(OUT, "sample_le_cornell_sphere.npz"), u1=u1, u2=u2, upick=upick, org=org, dir=d, nrm=nl,
                        le=le, pdf_pos=pp, pdf_dir=pd)
    osc.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "materials":
        materials_and_lights()
    else:
        main()
