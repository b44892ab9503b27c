"""ctypes mirror of include/tkhip.h and include/tkhost.h, and the library loaders.

The product path has no CPU fallback: if libtkhip.so is missing or cannot be
loaded, load_tkhip() raises and every operation built on it fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)
c_uint16_p = C.POINTER(C.c_uint16)
c_uint64_p = C.POINTER(C.c_uint64)

TK_DIFFUSE, TK_MIRROR, TK_GLASS = 1, 2, 4
TKHIP_MAX_K = 448


class TkTriangle(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("v1", C.c_float * 3), ("v2", C.c_float * 3)]


class TkSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float)]


class TkMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("R", C.c_float * 3), ("T", C.c_float * 3),
                ("etaA", C.c_float), ("etaB", C.c_float)]


class TkLight(C.Structure):
    _fields_ = [("Le", C.c_float * 3), ("two_sided", C.c_int32), ("object", C.c_int32)]


class TkObject(C.Structure):
    _fields_ = [("kind", C.c_int32), ("first", C.c_int32), ("count", C.c_int32),
                ("material", C.c_int32), ("light", C.c_int32)]


class TkSppmParam(C.Structure):
    _fields_ = [("numRecurse", C.c_int32), ("globalSize", C.c_int32), ("globalSample", C.c_int32),
                ("causticSize", C.c_int32), ("causticSample", C.c_int32), ("radius2", C.c_float)]


class TkSPPMPixel(C.Structure):
    _fields_ = [("Ld", C.c_float * 3), ("causticFlux", C.c_float * 3), ("globalFlux", C.c_float * 3),
                ("causticPhotons", C.c_float), ("globalPhotons", C.c_float),
                ("causticRadius2", C.c_float), ("globalRadius2", C.c_float)]


class TkIterStats(C.Structure):
    _fields_ = [("global_paths", C.c_int64), ("caustic_paths", C.c_int64),
                ("global_stored", C.c_int64), ("caustic_stored", C.c_int64),
                ("shoot_global", C.c_int64), ("shoot_caustic", C.c_int64),
                ("rays", C.c_int64), ("gathers_global", C.c_int64), ("gathers_caustic", C.c_int64),
                ("gather_photons", C.c_int64), ("overflow_global", C.c_int64), ("overflow_caustic", C.c_int64),
                ("group_staged", C.c_int64),
                ("ms_photon", C.c_float), ("ms_build", C.c_float), ("ms_camera", C.c_float),
                ("ms_gather", C.c_float), ("ms_gather_kernel", C.c_float), ("ms_total", C.c_float),
                ("rays_camera", C.c_int64),
                ("node_visits_camera", C.c_int64), ("prim_tests_camera", C.c_int64),
                ("node_visits_light", C.c_int64), ("prim_tests_light", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class TkGatherTotals(C.Structure):
    _fields_ = [("launches", C.c_int64), ("queries", C.c_int64), ("queries_handed_on", C.c_int64),
                ("photons_found", C.c_int64), ("photons_staged", C.c_int64), ("photons_found_k", C.c_int64 * 2),
                ("queries_handed_on_k", C.c_int64 * 2), ("ms_group_kernel", C.c_double),
                ("ms_group_side_by_side", C.c_double), ("steps_side_by_side", C.c_int64)]


class TkTracerConfig(C.Structure):
    _fields_ = [("type", C.c_char * 16), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("num_threads", C.c_int32), ("russian_roulette", C.c_float), ("sppm", TkSppmParam)]


class TkCameraConfig(C.Structure):
    _fields_ = [("fov_degrees", C.c_float), ("near_clip", C.c_float), ("far_clip", C.c_float),
                ("theta", C.c_float), ("phi", C.c_float), ("r", C.c_float)]


# name -> (restype, argtypes); every symbol include/tkhip.h declares
TKHIP_SYMBOLS = {
    "tkhip_last_error": (C.c_char_p, []),
    "tkhip_version": (C.c_char_p, []),
    "tkhip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "tkhip_destroy": (None, [C.c_void_p]),
    "tkhip_set_scene": (C.c_int, [C.c_void_p, C.POINTER(TkObject), C.c_int, C.POINTER(TkTriangle), C.c_int,
                                  C.POINTER(TkSphere), C.c_int, C.POINTER(TkMaterial), C.c_int,
                                  C.POINTER(TkLight), C.c_int]),
    "tkhip_set_camera": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int]),
    "tkhip_sppm_begin": (C.c_int, [C.c_void_p, C.POINTER(TkSppmParam), C.c_int, C.c_float, C.c_uint64,
                                   C.c_int, C.c_int]),
    "tkhip_sppm_photon_pass": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_sppm_export_local_photons": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_int64, c_int64_p, c_int64_p]),
    "tkhip_sppm_set_map_photons": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_int64, C.c_int64]),
    "tkhip_sppm_build_maps": (C.c_int, [C.c_void_p, C.POINTER(TkIterStats)]),
    "tkhip_sppm_set_tree": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_sppm_camera_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_sppm_photon_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_sppm_gather_totals": (C.c_int, [C.c_void_p, C.POINTER(TkGatherTotals), C.c_int]),
    "tkhip_sppm_camera_pass": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_sppm_camera_pass_ahead": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_sppm_iteration": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_sppm_read_pixels": (C.c_int, [C.c_void_p, C.POINTER(TkSPPMPixel)]),
    "tkhip_sppm_export_planes": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tkhip_sppm_import_planes": (C.c_int, [C.c_void_p, C.c_void_p]),
    "tkhip_sppm_resolve_rgb8": (C.c_int, [C.c_void_p, C.c_int, c_uint8_p]),
    "tkhip_set_bvh_builder": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_bvh_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_float_p]),
    "tkhip_hsppm_begin": (C.c_int, [C.c_void_p, C.POINTER(TkSppmParam), C.c_int, C.c_int, C.c_float, C.c_uint64]),
    "tkhip_hsppm_iteration": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_hsppm_resolve_rgb8": (C.c_int, [C.c_void_p, c_uint8_p]),
    "tkhip_pt_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_uint64]),
    "tkhip_pt_render_samples": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_pt_resolve_rgb8": (C.c_int, [C.c_void_p, c_uint8_p]),
    "tkhip_device_count": (C.c_int, []),
    "tkhip_group_create": (C.c_int, [c_int32_p, C.c_int, C.POINTER(C.c_void_p)]),
    "tkhip_group_destroy": (None, [C.c_void_p]),
    "tkhip_group_size": (C.c_int, [C.c_void_p]),
    "tkhip_group_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "tkhip_group_set_scene": (C.c_int, [C.c_void_p, C.POINTER(TkObject), C.c_int, C.POINTER(TkTriangle), C.c_int,
                                        C.POINTER(TkSphere), C.c_int, C.POINTER(TkMaterial), C.c_int,
                                        C.POINTER(TkLight), C.c_int]),
    "tkhip_group_set_camera": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int, C.c_int]),
    "tkhip_group_sppm_begin": (C.c_int, [C.c_void_p, C.POINTER(TkSppmParam), C.c_int, C.c_float, C.c_uint64]),
    "tkhip_group_sppm_iteration": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(TkIterStats)]),
    "tkhip_group_sppm_merge": (C.c_int, [C.c_void_p]),
    "tkhip_sppm_set_preview": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_sppm_preview": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), c_int32_p]),
    "tkhip_sppm_read_preview": (C.c_int, [C.c_void_p, C.c_void_p, c_int32_p]),
    "tkhip_sppm_counters": (C.c_int, [C.c_void_p, c_int64_p, c_int64_p, c_int32_p]),
    "tkhip_map_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "tkhip_map_destroy": (None, [C.c_void_p]),
    "tkhip_map_store": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_uint8_p, c_uint8_p, C.c_int64]),
    "tkhip_map_set_tree": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_map_build": (C.c_int, [C.c_void_p]),
    "tkhip_map_tree_info": (C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "tkhip_map_gather": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_float_p,
                                   c_float_p, c_float_p, c_int32_p, c_float_p, c_int32_p, C.c_int]),
    "tkhip_map_gather_static": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tkhip_trace_rays": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_int32_p,
                                   c_float_p, c_float_p, c_float_p, c_float_p]),
    "tkhip_pixel_update": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_float_p,
                                     c_int32_p, c_float_p]),
    "tkhip_radical_inverse": (C.c_int, [C.c_void_p, C.c_int64, c_int32_p, c_uint64_p, c_float_p]),
    "tkhip_camera_rays": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p]),
    "tkhip_sample_f": (C.c_int, [C.c_void_p, C.POINTER(TkMaterial), C.c_int, C.c_int64, c_float_p, c_float_p, c_float_p,
                                 c_float_p, c_float_p, c_float_p]),
    "tkhip_sample_le": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                  c_float_p, c_float_p, c_float_p, c_float_p]),
    "tkhip_sppm_set_jitter": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_sppm_exchange_words": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "tkhip_sppm_export_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tkhip_sppm_import_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "tkhip_sppm_set_counting": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_stratified_samples": (C.c_int, [C.c_void_p, C.c_int64, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                           C.c_int, c_uint8_p, C.c_int, C.c_uint64, c_float_p]),
    "tkhip_spawn_rays": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                   c_float_p, c_float_p, c_float_p]),
    "tkhip_sample_one_light": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_float_p, c_float_p, c_int32_p, c_float_p]),
    "tkhip_set_wide_walk": (C.c_int, [C.c_void_p, C.c_int]),
    "tkhip_encode_directions": (C.c_int, [C.c_void_p, C.c_int64, c_float_p, c_uint8_p, c_uint8_p]),
    "tkhip_sppm_read_photons": (C.c_int, [C.c_void_p, C.c_int, c_float_p, c_float_p, c_uint8_p, c_uint8_p,
                                          c_int64_p]),
}

TKHOST_SYMBOLS = {
    "tkhost_last_error": (C.c_char_p, []),
    "tkhost_load_scene": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "tkhost_free_scene": (None, [C.c_void_p]),
    "tkhost_make_synthetic": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "tkhost_objects": (C.POINTER(TkObject), [C.c_void_p, C.POINTER(C.c_int)]),
    "tkhost_triangles": (C.POINTER(TkTriangle), [C.c_void_p, C.POINTER(C.c_int)]),
    "tkhost_spheres": (C.POINTER(TkSphere), [C.c_void_p, C.POINTER(C.c_int)]),
    "tkhost_materials": (C.POINTER(TkMaterial), [C.c_void_p, C.POINTER(C.c_int)]),
    "tkhost_lights": (C.POINTER(TkLight), [C.c_void_p, C.POINTER(C.c_int)]),
    "tkhost_tracer_config": (None, [C.c_void_p, C.POINTER(TkTracerConfig)]),
    "tkhost_camera_config": (None, [C.c_void_p, C.POINTER(TkCameraConfig)]),
    "tkhost_image_config": (None, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                   C.c_char_p, C.c_size_t]),
    "tkhost_camera_matrices": (None, [C.POINTER(TkCameraConfig), C.c_int, C.c_int, c_float_p, c_float_p]),
    "tkhost_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint64,
                                C.POINTER(C.c_double)]),
    "tkhost_render_windowed": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.c_int,
                                         C.c_char_p, C.c_void_p]),
    "tkhost_gpu_count": (C.c_int, [C.c_void_p]),
    "tkhost_set_gpu_count": (None, [C.c_void_p, C.c_int]),
    "tkhost_write_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, c_uint8_p]),
}

_libs = {}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def _load(name, table):
    if name not in _libs:
        path = os.path.join(LIB_DIR, name)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` at the repository root "
                               "(this package has no CPU fallback)")
        _libs[name] = _bind(C.CDLL(path, mode=C.RTLD_GLOBAL), table)
    return _libs[name]


def load_tkhip():
    return _load("libtkhip.so", TKHIP_SYMBOLS)


def load_tkhost():
    load_tkhip()  # libtkhost.so links against it
    return _load("libtkhost.so", TKHOST_SYMBOLS)


class TkError(RuntimeError):
    pass


def check(lib, rc, what):
    if rc != 0:
        msg = lib.tkhip_last_error() if hasattr(lib, "tkhip_last_error") else b""
        raise TkError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")
