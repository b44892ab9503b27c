"""Python host side over the C ABI (libtkhip.so).

`SppmTracer` mirrors the reference tracer interface (src/tracer/TkSppmTracer.h:48-52,
src/core/TkRayTracer.h:46-59): construct from the <tracer> block, setScene,
setCamera, render(filename).  `Context` and `PhotonMap` expose the kernel-level
entry points for parity tests.  Nothing here computes on the CPU: every
operation is a call into the HIP library and raises if that fails.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import check


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_abi.c_float_p)


def _ip(a):
    return a.ctypes.data_as(_abi.c_int32_p)


def _bp(a):
    return a.ctypes.data_as(_abi.c_uint8_p)


class Context:
    """One GPU context (tkhip_create)."""

    def __init__(self, device=0):
        self.lib = _abi.load_tkhip()
        h = C.c_void_p()
        check(self.lib, self.lib.tkhip_create(device, C.byref(h)), "tkhip_create")
        self.h = h
        self.device = device
        self.width = self.height = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.tkhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bvh_builder(self, mode):
        """0: GPU from 4096 primitives on, host threads below; 1: host threads; 2: GPU.  Before set_scene."""
        check(self.lib, self.lib.tkhip_set_bvh_builder(self.h, mode), "tkhip_set_bvh_builder")

    def bvh_info(self):
        n, d, dev, ms = C.c_int32(), C.c_int32(), C.c_int32(), C.c_float()
        check(self.lib, self.lib.tkhip_bvh_info(self.h, C.byref(n), C.byref(d), C.byref(dev), C.byref(ms)), "tkhip_bvh_info")
        return {"nodes": n.value, "depth": d.value, "on_device": bool(dev.value), "ms": ms.value}

    def set_scene(self, scene):
        check(self.lib, self.lib.tkhip_set_scene(self.h, scene.objects, scene.n_objects, scene.triangles,
                                                 scene.n_triangles, scene.spheres, scene.n_spheres,
                                                 scene.materials, scene.n_materials, scene.lights, scene.n_lights),
              "tkhip_set_scene")
        self.scene = scene

    def set_camera(self, scene, width=None, height=None):
        width = width or scene.width
        height = height or scene.height
        c2w, r2c = scene.camera_matrices(width, height)
        check(self.lib, self.lib.tkhip_set_camera(self.h, _fp(c2w), _fp(r2c), width, height), "tkhip_set_camera")
        self.width, self.height = width, height

    # ---- kernel-level entry points
    def trace_rays(self, org, direction, tmax=None):
        org, direction = _f32(org), _f32(direction)
        n = org.shape[0]
        tm = _f32(tmax) if tmax is not None else None
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        p = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        perr = np.empty((n, 3), np.float32)
        check(self.lib, self.lib.tkhip_trace_rays(self.h, n, _fp(org), _fp(direction), _fp(tm) if tm is not None else None,
                                                  _ip(prim), _fp(t), _fp(p), _fp(nrm), _fp(perr)), "tkhip_trace_rays")
        return prim, t, p, nrm, perr

    def camera_rays(self, raster):
        raster = _f32(raster)
        n = raster.shape[0]
        org = np.empty((n, 3), np.float32)
        d = np.empty((n, 3), np.float32)
        check(self.lib, self.lib.tkhip_camera_rays(self.h, n, _fp(raster), _fp(org), _fp(d)), "tkhip_camera_rays")
        return org, d

    def sample_f(self, material, mode, wo, nrm, u):
        """Material::sample_f for n samples on one TkMaterial (mode 0 radiance, 1 importance)."""
        wo, nrm, u = _f32(wo), _f32(nrm), _f32(u)
        n = wo.shape[0]
        wi = np.empty((n, 3), np.float32)
        f = np.empty((n, 3), np.float32)
        pdf = np.empty(n, np.float32)
        check(self.lib, self.lib.tkhip_sample_f(self.h, C.byref(material), mode, n, _fp(wo), _fp(nrm), _fp(u), _fp(wi), _fp(f),
                                                _fp(pdf)), "tkhip_sample_f")
        return wi, f, pdf

    def sample_le(self, light, u1, u2, upick):
        """AreaLight::sample_Le on a light of the scene: (origin, direction, normal, Le, pdf_pos, pdf_dir)."""
        u1, u2, upick = _f32(u1), _f32(u2), _f32(upick)
        n = upick.shape[0]
        out = [np.empty((n, 3), np.float32) for _ in range(4)] + [np.empty(n, np.float32) for _ in range(2)]
        check(self.lib, self.lib.tkhip_sample_le(self.h, light, n, _fp(u1), _fp(u2), _fp(upick), *[_fp(o) for o in out]),
              "tkhip_sample_le")
        return tuple(out)

    def radical_inverse(self, dim, index):
        dim = np.ascontiguousarray(dim, np.int32)
        index = np.ascontiguousarray(index, np.uint64)
        out = np.empty(dim.shape[0], np.float32)
        check(self.lib, self.lib.tkhip_radical_inverse(self.h, dim.shape[0], _ip(dim),
                                                       index.ctypes.data_as(_abi.c_uint64_p), _fp(out)),
              "tkhip_radical_inverse")
        return out

    def stratified_samples(self, px, py, sample_idx, xs, ys, draws, jitter=False, seed=0):
        """StratifiedSampler draws (1 = get1D, 2 = get2D) for n (pixel, sample index) pairs."""
        arrs = [np.ascontiguousarray(a, np.int32) for a in (px, py, sample_idx, xs, ys)]
        draws = np.ascontiguousarray(draws, np.uint8)
        n = arrs[0].shape[0]
        out = np.empty((n, int(draws.sum())), np.float32)
        check(self.lib, self.lib.tkhip_stratified_samples(self.h, n, *[_ip(a) for a in arrs], draws.shape[0], _bp(draws),
                                                          1 if jitter else 0, seed, _fp(out)), "tkhip_stratified_samples")
        return out

    def spawn_rays(self, p, perr, nrm, w, target):
        """Intersection::spawnRay(w) origin, spawnRayTo(target) origin and direction."""
        ins = [_f32(a) for a in (p, perr, nrm, w, target)]
        n = ins[0].shape[0]
        outs = [np.empty((n, 3), np.float32) for _ in range(3)]
        check(self.lib, self.lib.tkhip_spawn_rays(self.h, n, *[_fp(a) for a in ins], *[_fp(o) for o in outs]),
              "tkhip_spawn_rays")
        return tuple(outs)

    def set_wide_walk(self, max_rays):
        check(self.lib, self.lib.tkhip_set_wide_walk(self.h, int(max_rays)), "tkhip_set_wide_walk")

    def sample_one_light(self, org, direction, u):
        """Scene::uniformSampleOneLight at the first hit of each ray; u: the four sampler numbers per ray."""
        org, direction, u = _f32(org), _f32(direction), _f32(u)
        n = org.shape[0]
        prim = np.empty(n, np.int32)
        ld = np.empty((n, 3), np.float32)
        check(self.lib, self.lib.tkhip_sample_one_light(self.h, n, _fp(org), _fp(direction), _fp(u), _ip(prim), _fp(ld)),
              "tkhip_sample_one_light")
        return prim, ld

    def encode_directions(self, wo):
        wo = _f32(wo)
        n = wo.shape[0]
        th, ph = np.empty(n, np.uint8), np.empty(n, np.uint8)
        check(self.lib, self.lib.tkhip_encode_directions(self.h, n, _fp(wo), _bp(th), _bp(ph)), "tkhip_encode_directions")
        return th, ph

    def pixel_update(self, flux, photons, radius2, lindir, found, radius2_ret):
        flux, photons, radius2 = _f32(flux).copy(), _f32(photons).copy(), _f32(radius2).copy()
        lindir, radius2_ret = _f32(lindir), _f32(radius2_ret)
        found = np.ascontiguousarray(found, np.int32)
        check(self.lib, self.lib.tkhip_pixel_update(self.h, photons.shape[0], _fp(flux), _fp(photons), _fp(radius2),
                                                    _fp(lindir), _ip(found), _fp(radius2_ret)), "tkhip_pixel_update")
        return flux, photons, radius2


class PhotonMap:
    """PhotonMap(size, sampleCount) of the reference (src/tracer/Photonmap.h:31-40) on the device."""

    def __init__(self, ctx, capacity, k):
        self.ctx, self.lib, self.k = ctx, ctx.lib, k
        h = C.c_void_p()
        check(self.lib, self.lib.tkhip_map_create(ctx.h, capacity, k, C.byref(h)), "tkhip_map_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.tkhip_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def store(self, pos, power, theta, phi):
        pos, power = _f32(pos), _f32(power)
        theta = np.ascontiguousarray(theta, np.uint8)
        phi = np.ascontiguousarray(phi, np.uint8)
        check(self.lib, self.lib.tkhip_map_store(self.h, _fp(pos), _fp(power), _bp(theta), _bp(phi), pos.shape[0]),
              "tkhip_map_store")

    def set_tree(self, mode):
        """0: one sort along a Hilbert curve (default); 1: median splits, one sort per level."""
        check(self.lib, self.lib.tkhip_map_set_tree(self.h, mode), "tkhip_map_set_tree")

    def build(self):
        check(self.lib, self.lib.tkhip_map_build(self.h), "tkhip_map_build")

    def tree_info(self, n_photons=0):
        nl, cap, depth = C.c_int32(), C.c_int32(), C.c_int32()
        perm = np.empty(n_photons, np.int32) if n_photons else None
        check(self.lib, self.lib.tkhip_map_tree_info(self.h, C.byref(nl), C.byref(cap), C.byref(depth),
                                                     _ip(perm) if perm is not None else None), "tkhip_map_tree_info")
        return nl.value, cap.value, depth.value, perm

    def gather(self, p, n, wo, R, radius2, mode=0, want_idx=True):
        p, n, wo, R, radius2 = _f32(p), _f32(n), _f32(wo), _f32(R), _f32(radius2)
        nq = p.shape[0]
        flux = np.empty((nq, 3), np.float32)
        found = np.empty(nq, np.int32)
        r2 = np.empty(nq, np.float32)
        idx = np.empty((nq, self.k), np.int32) if want_idx else None
        check(self.lib, self.lib.tkhip_map_gather(self.h, nq, _fp(p), _fp(n), _fp(wo), _fp(R), _fp(radius2), _fp(flux),
                                                  _ip(found), _fp(r2), _ip(idx) if idx is not None else None, mode),
              "tkhip_map_gather")
        if idx is not None:
            # ascending, -1 padding last (the kernels emit the set in no particular order)
            big = np.where(idx < 0, np.iinfo(np.int32).max, idx)
            big.sort(axis=1)
            idx = np.where(big == np.iinfo(np.int32).max, -1, big).astype(np.int32)
        return flux, found, r2, idx


class SppmTracer:
    """SppmTracer(spp, maxDepth, numThreads, russianRoulette, param): src/tracer/TkSppmTracer.h:48."""

    def __init__(self, spp, max_depth, num_threads, russian_roulette, param, device=0, seed=0x5EED, rank=0, world=1):
        self.spp, self.max_depth, self.num_threads = spp, max_depth, num_threads
        self.rr, self.param = russian_roulette, param
        self.device, self.seed, self.rank, self.world = device, seed, rank, world
        self.ctx = None
        self.scene = None
        self.width = self.height = 0
        self.cur_recurse = 0
        self.num_recurse = 0
        self.stats = []
        self.state = "INIT"
        self.bvh_builder = 0   # Context.set_bvh_builder mode used by startRaytracing

    @classmethod
    def from_scene(cls, scene, **kw):
        t = scene.tracer
        over = {k: kw.pop(k) for k in list(kw) if k in ("numRecurse", "globalSize", "globalSample", "causticSize",
                                                         "causticSample", "radius2")}
        return cls(t.spp, t.max_depth, t.num_threads, t.russian_roulette, scene.sppm_param(**over), **kw)

    def setScene(self, scene):
        if self.state != "INIT":
            return
        self.scene = scene

    def setCamera(self, width=None, height=None):
        if self.state not in ("INIT", "READY"):
            return
        self.width = width or self.scene.width
        self.height = height or self.scene.height
        self.state = "READY"

    def startRaytracing(self):
        if self.state != "READY":
            return
        if self.ctx:
            self.ctx.close()
        self.ctx = Context(self.device)
        self.ctx.set_bvh_builder(self.bvh_builder)
        self.ctx.set_scene(self.scene)
        self.ctx.set_camera(self.scene, self.width, self.height)
        lib = self.ctx.lib
        check(lib, lib.tkhip_sppm_begin(self.ctx.h, C.byref(self.param), self.max_depth, self.rr, self.seed,
                                        self.rank, self.world), "tkhip_sppm_begin")
        nr = C.c_int32()
        check(lib, lib.tkhip_sppm_counters(self.ctx.h, None, None, C.byref(nr)), "tkhip_sppm_counters")
        self.num_recurse = nr.value
        self.cur_recurse = 0
        self.stats = []
        self.state = "RENDERING"

    def set_jitter(self, on):
        """False: the camera sampler takes stratum centres (StratifiedSampler with jitter = false)."""
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_set_jitter(self.ctx.h, 1 if on else 0), "tkhip_sppm_set_jitter")

    # phases of one iteration (world > 1 drives them separately around the exchange)
    def photon_pass(self, it, stats=None):
        lib = self.ctx.lib
        check(lib, lib.tkhip_sppm_photon_pass(self.ctx.h, it, C.byref(stats) if stats is not None else None),
              "tkhip_sppm_photon_pass")

    def build_maps(self, stats=None):
        lib = self.ctx.lib
        check(lib, lib.tkhip_sppm_build_maps(self.ctx.h, C.byref(stats) if stats is not None else None),
              "tkhip_sppm_build_maps")

    def camera_pass(self, it, stats=None, next_ahead=False):
        """next_ahead: the camera paths of iteration it + 1 are started before this iteration's gathers
        (tkhip_sppm_camera_pass_ahead); the caller then has to run iteration it + 1."""
        lib = self.ctx.lib
        fn = lib.tkhip_sppm_camera_pass_ahead if next_ahead else lib.tkhip_sppm_camera_pass
        check(lib, fn(self.ctx.h, it, C.byref(stats) if stats is not None else None),
              "tkhip_sppm_camera_pass_ahead" if next_ahead else "tkhip_sppm_camera_pass")

    def iteration(self, it=None, with_stats=True):
        """One Generate -> Rendering round of the state machine (src/tracer/TkSppmTracer.cpp:454-541)."""
        it = self.cur_recurse if it is None else it
        st = _abi.TkIterStats() if with_stats else None
        lib = self.ctx.lib
        check(lib, lib.tkhip_sppm_iteration(self.ctx.h, it, C.byref(st) if st is not None else None),
              "tkhip_sppm_iteration")
        self.cur_recurse = it + 1
        if st is not None:
            self.stats.append(st)
        if self.cur_recurse >= self.num_recurse:
            self.state = "DONE"
        return st

    def rendering(self):
        if self.state == "RENDERING":
            self.iteration()

    def render(self, filename=None):
        self.startRaytracing()
        while self.state == "RENDERING":
            self.rendering()
        img = self.resolve()
        if filename:
            self.saveImage(filename, img)
        return img

    def pixels(self):
        n = self.width * self.height
        buf = (_abi.TkSPPMPixel * n)()
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_read_pixels(self.ctx.h, buf), "tkhip_sppm_read_pixels")
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 13).copy()

    def resolve(self, iterations=None):
        out = np.empty((self.height, self.width, 3), np.uint8)
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_resolve_rgb8(self.ctx.h, iterations or self.num_recurse, _bp(out)),
              "tkhip_sppm_resolve_rgb8")
        return out

    def set_preview(self, every_n):
        """Device-resident RGBA8 frame buffer refreshed after every every_n-th iteration (0: never)."""
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_set_preview(self.ctx.h, every_n), "tkhip_sppm_set_preview")

    def preview(self, refresh=False):
        """(frame buffer as (H, W, 4) uint8 RGBA in SPPMPixel row order, iteration it shows)."""
        buf = np.empty((self.height, self.width), np.uint32)
        it = C.c_int32(-1)
        if refresh:
            check(self.ctx.lib, self.ctx.lib.tkhip_sppm_preview(self.ctx.h, 1, None, None), "tkhip_sppm_preview")
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_read_preview(self.ctx.h, buf.ctypes.data, C.byref(it)),
              "tkhip_sppm_read_preview")
        return buf.view(np.uint8).reshape(self.height, self.width, 4), it.value

    def saveImage(self, filename, img=None):
        img = self.resolve() if img is None else img
        with open(filename, "wb") as f:
            f.write(b"P6\n%d %d\n255\n" % (self.width, self.height))
            f.write(img.tobytes())

    def counters(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_counters(self.ctx.h, C.byref(a), C.byref(b), C.byref(c)),
              "tkhip_sppm_counters")
        return a.value, b.value, c.value

    def photons(self, which):
        cap = self.param.globalSize if which == 0 else max(self.param.causticSize, 1)
        pos = np.empty((cap, 3), np.float32)
        power = np.empty((cap, 3), np.float32)
        theta = np.empty(cap, np.uint8)
        phi = np.empty(cap, np.uint8)
        n = C.c_int64()
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_read_photons(self.ctx.h, which, _fp(pos), _fp(power), _bp(theta),
                                                                 _bp(phi), C.byref(n)), "tkhip_sppm_read_photons")
        n = n.value
        return pos[:n], power[:n], theta[:n], phi[:n]

    def close(self):
        if self.ctx:
            self.ctx.close()
            self.ctx = None


class PathTracer:
    """tk::PathTracer on the GPU (tkhip_pt_*): spp samples per pixel into the box-filter film."""

    def __init__(self, scene, spp, max_depth, rr, width, height, device=0, seed=0x5EED):
        self.ctx = Context(device)
        self.ctx.set_scene(scene)
        self.ctx.set_camera(scene, width, height)
        self.width, self.height, self.spp = width, height, spp
        check(self.ctx.lib, self.ctx.lib.tkhip_pt_begin(self.ctx.h, spp, max_depth, rr, seed), "tkhip_pt_begin")

    def set_jitter(self, on):
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_set_jitter(self.ctx.h, 1 if on else 0), "tkhip_sppm_set_jitter")

    def render_samples(self, count):
        st = _abi.TkIterStats()
        check(self.ctx.lib, self.ctx.lib.tkhip_pt_render_samples(self.ctx.h, count, C.byref(st)), "tkhip_pt_render_samples")
        return st

    def film(self):
        """(contribSum (H*W, 3), filterWeightSum (H*W,))"""
        n = self.width * self.height
        buf = (_abi.TkSPPMPixel * n)()
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_read_pixels(self.ctx.h, buf), "tkhip_sppm_read_pixels")
        px = np.frombuffer(buf, dtype=np.float32).reshape(n, 13)
        return px[:, 0:3].copy(), px[:, 9].copy()

    def resolve(self):
        out = np.empty((self.height, self.width, 3), np.uint8)
        check(self.ctx.lib, self.ctx.lib.tkhip_pt_resolve_rgb8(self.ctx.h, _bp(out)), "tkhip_pt_resolve_rgb8")
        return out

    def close(self):
        if self.ctx:
            self.ctx.close()
            self.ctx = None


class HashGridSppm:
    """The hash-grid SPPM variant (tkhip_hsppm_*): visible point per pixel, photons deposited through a
    spatial hash grid.  param.globalSize = light paths per iteration, param.radius2 = the initial radius."""

    def __init__(self, scene, param, spp, max_depth, rr, width, height, device=0, seed=0x5EED):
        self.ctx = Context(device)
        self.ctx.set_scene(scene)
        self.ctx.set_camera(scene, width, height)
        self.width, self.height, self.param = width, height, param
        lib = self.ctx.lib
        check(lib, lib.tkhip_hsppm_begin(self.ctx.h, C.byref(param), spp, max_depth, rr, seed), "tkhip_hsppm_begin")
        self.cur = 0

    def set_jitter(self, on):
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_set_jitter(self.ctx.h, 1 if on else 0), "tkhip_sppm_set_jitter")

    def iteration(self, it=None):
        it = self.cur if it is None else it
        st = _abi.TkIterStats()
        check(self.ctx.lib, self.ctx.lib.tkhip_hsppm_iteration(self.ctx.h, it, C.byref(st)), "tkhip_hsppm_iteration")
        self.cur = it + 1
        return st

    def pixels(self):
        n = self.width * self.height
        buf = (_abi.TkSPPMPixel * n)()
        check(self.ctx.lib, self.ctx.lib.tkhip_sppm_read_pixels(self.ctx.h, buf), "tkhip_sppm_read_pixels")
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 13).copy()

    def resolve(self):
        out = np.empty((self.height, self.width, 3), np.uint8)
        check(self.ctx.lib, self.ctx.lib.tkhip_hsppm_resolve_rgb8(self.ctx.h, _bp(out)), "tkhip_hsppm_resolve_rgb8")
        return out

    def close(self):
        if self.ctx:
            self.ctx.close()
            self.ctx = None


class SppmGroup:
    """One SPPM render over several GPUs of this process (tkhip_group_*): one context and one host thread per
    rank inside the library, photons exchanged with peer copies.  `devices` may repeat a GPU."""

    def __init__(self, scene, devices, param, max_depth, rr, width, height, seed=0x5EED):
        self.lib = _abi.load_tkhip()
        lib = self.lib
        dev = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        check(lib, lib.tkhip_group_create(dev, len(devices), C.byref(h)), "tkhip_group_create")
        self.h, self.world, self.width, self.height, self.scene = h, len(devices), width, height, scene
        check(lib, lib.tkhip_group_set_scene(h, scene.objects, scene.n_objects, scene.triangles, scene.n_triangles,
                                             scene.spheres, scene.n_spheres, scene.materials, scene.n_materials,
                                             scene.lights, scene.n_lights), "tkhip_group_set_scene")
        c2w, r2c = scene.camera_matrices(width, height)
        check(lib, lib.tkhip_group_set_camera(h, _fp(c2w), _fp(r2c), width, height), "tkhip_group_set_camera")
        self.param = param
        check(lib, lib.tkhip_group_sppm_begin(h, C.byref(param), max_depth, rr, seed), "tkhip_group_sppm_begin")
        self.ctx0 = lib.tkhip_group_ctx(h, 0)

    def iteration(self, it, begin_next=False, with_stats=False):
        stats = (_abi.TkIterStats * self.world)() if with_stats else None
        check(self.lib, self.lib.tkhip_group_sppm_iteration(self.h, it, int(begin_next), stats),
              "tkhip_group_sppm_iteration")
        return stats

    def merge(self):
        check(self.lib, self.lib.tkhip_group_sppm_merge(self.h), "tkhip_group_sppm_merge")

    def pixels(self):
        """The merged SPPMPixel array, (width*height, 13) float32."""
        self.merge()
        n = self.width * self.height
        buf = (_abi.TkSPPMPixel * n)()
        check(self.lib, self.lib.tkhip_sppm_read_pixels(self.ctx0, buf), "tkhip_sppm_read_pixels")
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 13).copy()

    def resolve(self, iterations):
        self.merge()
        out = np.empty((self.height, self.width, 3), np.uint8)
        check(self.lib, self.lib.tkhip_sppm_resolve_rgb8(self.ctx0, iterations, _bp(out)), "tkhip_sppm_resolve_rgb8")
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.tkhip_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
