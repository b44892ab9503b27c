"""Scene loading through the C++ loader (libtkhost.so).

A Scene owns the loaded tkhost_scene and exposes the POD arrays (as ctypes
pointers + counts) that tkhip_set_scene and the CPU oracle take, the tracer
block and the camera matrices.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def asset_scene(name):
    return os.path.join(ASSETS, "scenes", name)


class Scene:
    def __init__(self, handle):
        self._lib = _abi.load_tkhost()
        self._h = handle
        n = C.c_int()
        self.objects = self._lib.tkhost_objects(self._h, C.byref(n)); self.n_objects = n.value
        self.triangles = self._lib.tkhost_triangles(self._h, C.byref(n)); self.n_triangles = n.value
        self.spheres = self._lib.tkhost_spheres(self._h, C.byref(n)); self.n_spheres = n.value
        self.materials = self._lib.tkhost_materials(self._h, C.byref(n)); self.n_materials = n.value
        self.lights = self._lib.tkhost_lights(self._h, C.byref(n)); self.n_lights = n.value
        self.tracer = _abi.TkTracerConfig()
        self._lib.tkhost_tracer_config(self._h, C.byref(self.tracer))
        self.camera = _abi.TkCameraConfig()
        self._lib.tkhost_camera_config(self._h, C.byref(self.camera))
        w, h, win = C.c_int(), C.c_int(), C.c_int()
        buf = C.create_string_buffer(512)
        self._lib.tkhost_image_config(self._h, C.byref(w), C.byref(h), C.byref(win), buf, 512)
        self.width, self.height, self.windows = w.value, h.value, win.value
        self.outfilename = buf.value.decode()

    @classmethod
    def load(cls, path):
        lib = _abi.load_tkhost()
        h = C.c_void_p()
        if lib.tkhost_load_scene(os.fsencode(path), C.byref(h)) != 0:
            raise _abi.TkError(f"load_scene({path}): {lib.tkhost_last_error().decode(errors='replace')}")
        return cls(h)

    @classmethod
    def synthetic(cls, quads, seed=0x5EED, model_dir=None):
        lib = _abi.load_tkhost()
        h = C.c_void_p()
        model_dir = model_dir or os.path.join(ASSETS, "models")
        if lib.tkhost_make_synthetic(os.fsencode(model_dir), quads, seed, C.byref(h)) != 0:
            raise _abi.TkError(f"make_synthetic: {lib.tkhost_last_error().decode(errors='replace')}")
        return cls(h)

    def close(self):
        if self._h:
            self._lib.tkhost_free_scene(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def camera_matrices(self, width=None, height=None):
        width = width or self.width
        height = height or self.height
        c2w = (C.c_float * 16)()
        r2c = (C.c_float * 16)()
        self._lib.tkhost_camera_matrices(C.byref(self.camera), width, height, c2w, r2c)
        return np.array(c2w, dtype=np.float32), np.array(r2c, dtype=np.float32)

    def triangles_np(self):
        """(n,3,3) float32 copy of the world-space triangle vertices."""
        if self.n_triangles == 0:
            return np.zeros((0, 3, 3), np.float32)
        a = np.ctypeslib.as_array(C.cast(self.triangles, C.POINTER(C.c_float)), shape=(self.n_triangles, 3, 3))
        return a.copy()

    def sppm_param(self, **override):
        p = _abi.TkSppmParam()
        C.memmove(C.byref(p), C.byref(self.tracer.sppm), C.sizeof(p))
        for k, v in override.items():
            setattr(p, k, v)
        return p
