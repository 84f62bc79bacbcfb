"""Python face of the host-side callers (include/rt_host.h): Scene.LoadScene, camera uniforms, Renderer frame
conversion, BVH build.  Thin ctypes calls into librt_core.so — the work is done by the C++ classes in
raytracing_amd/csrc/host/ that mirror the reference's Scene / Model / Light / Camera / Renderer."""
import ctypes as C

import numpy as np

from . import _lib
from .core import RtError

HOST_EXPORTS = ["rth_last_error", "rth_load_scene", "rth_load_models", "rth_scene_buffer", "rth_load_timings",
                "rth_build_bvh", "rth_free", "rth_set_camera", "rth_orbit_camera", "rth_frame_rgba8", "rth_decode_png", "rth_decode_image",
                "rth_multi_create", "rth_multi_destroy", "rth_multi_size", "rth_multi_ctx", "rth_multi_load_scene", "rth_multi_load_models",
                "rth_multi_set_int", "rth_multi_resize", "rth_multi_set_camera", "rth_multi_render_frame", "rth_multi_used_rccl", "rth_partition"]

_bound = False


def _L():
    global _bound
    lib = _lib.load()
    if not _bound:
        vp, u32, u64, i32, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
        lib.rth_last_error.restype = C.c_char_p
        lib.rth_last_error.argtypes = []
        lib.rth_load_scene.restype = i32
        lib.rth_load_scene.argtypes = [vp, C.c_char_p, i32]
        lib.rth_load_models.restype = i32
        lib.rth_load_models.argtypes = [vp, C.POINTER(C.c_char_p), i32, C.POINTER(f), i32]
        lib.rth_scene_buffer.restype = i32
        lib.rth_scene_buffer.argtypes = [u32, C.POINTER(vp), C.POINTER(u64)]
        lib.rth_load_timings.restype = i32
        lib.rth_load_timings.argtypes = [C.POINTER(C.c_double)]
        lib.rth_build_bvh.restype = i32
        lib.rth_build_bvh.argtypes = [vp, u64, C.POINTER(vp), C.POINTER(u64)]
        lib.rth_free.restype = None
        lib.rth_free.argtypes = [vp]
        lib.rth_set_camera.restype = i32
        lib.rth_set_camera.argtypes = [vp, C.POINTER(f), C.POINTER(f), f, f, f]
        lib.rth_orbit_camera.restype = None
        lib.rth_orbit_camera.argtypes = [i32, C.POINTER(f), C.POINTER(f)]
        lib.rth_frame_rgba8.restype = i32
        lib.rth_frame_rgba8.argtypes = [vp, u32, u32, vp]
        for f in (lib.rth_decode_png, lib.rth_decode_image):
            f.restype = i32
            f.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)]
        lib.rth_multi_create.restype = i32
        lib.rth_multi_create.argtypes = [C.POINTER(vp), C.POINTER(i32), i32]
        lib.rth_multi_destroy.restype = None
        lib.rth_multi_destroy.argtypes = [vp]
        lib.rth_multi_size.restype = i32
        lib.rth_multi_size.argtypes = [vp]
        lib.rth_multi_ctx.restype = vp
        lib.rth_multi_ctx.argtypes = [vp, i32]
        lib.rth_multi_load_scene.restype = i32
        lib.rth_multi_load_scene.argtypes = [vp, C.c_char_p, i32]
        lib.rth_multi_load_models.restype = i32
        lib.rth_multi_load_models.argtypes = [vp, C.POINTER(C.c_char_p), i32, C.POINTER(C.c_float), i32]
        lib.rth_multi_set_int.restype = i32
        lib.rth_multi_set_int.argtypes = [vp, C.c_char_p, i32]
        lib.rth_multi_resize.restype = i32
        lib.rth_multi_resize.argtypes = [vp, u32, u32]
        lib.rth_multi_set_camera.restype = i32
        lib.rth_multi_set_camera.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float]
        lib.rth_multi_render_frame.restype = i32
        lib.rth_multi_render_frame.argtypes = [vp, u32, i32, i32]
        lib.rth_multi_used_rccl.restype = i32
        lib.rth_multi_used_rccl.argtypes = [vp]
        lib.rth_partition.restype = i32
        lib.rth_partition.argtypes = [u32, u32, u32, i32, i32, i32, C.POINTER(u32)]
        _bound = True
    return lib


def _chk(st):
    if st != 0:
        raise RtError(st, (_L().rth_last_error() or b"").decode())


def _h(cs):
    return cs._h if cs is not None else None


DEFAULT_CAMERA = ((0.0, 3.5, 14.0), (0.0, 0.0, -1.0))   # main.cpp:239, camera.h:21


class Scene:
    """Scene::LoadScene (Scene.cpp:15-62).  cs=None runs the loader without a GPU (buffers only)."""

    @staticmethod
    def LoadScene(cs, scene_index, models_root):
        _chk(_L().rth_load_scene(_h(cs), str(models_root).encode(), int(scene_index)))

    @staticmethod
    def LoadModels(cs, gltf_paths, lights):
        """lights: [(loc, dir, type, strength)] as in Scene.cpp's Light(...) calls."""
        arr = (C.c_char_p * len(gltf_paths))(*[str(p).encode() for p in gltf_paths])
        flat = []
        for loc, d, typ, strength in lights:
            flat += [float(v) for v in loc] + [float(v) for v in d] + [float(typ), float(strength)]
        la = (C.c_float * max(1, len(flat)))(*flat)
        _chk(_L().rth_load_models(_h(cs), arr, len(gltf_paths), la, len(lights)))

    @staticmethod
    def buffer(slot):
        """Host copy (bytes) of what the last load bound to `slot`."""
        p, n = C.c_void_p(), C.c_uint64()
        _chk(_L().rth_scene_buffer(slot, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value) if n.value else b""

    @staticmethod
    def buffer_view(slot):
        """The same without a copy: a uint8 numpy view of the loader's own array (valid until the next load)."""
        p, n = C.c_void_p(), C.c_uint64()
        _chk(_L().rth_scene_buffer(slot, C.byref(p), C.byref(n)))
        if not n.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,))

    @staticmethod
    def timings():
        t = (C.c_double * 3)()
        _chk(_L().rth_load_timings(t))
        return {"flatten_s": t[0], "bvh_build_s": t[1], "upload_s": t[2]}


def build_bvh(triangles):
    """Model::buildBVH + convertToGPU on a [n] array of 128-byte Triangle records -> bytes of 48-byte nodes."""
    buf = np.ascontiguousarray(triangles)
    assert buf.dtype.itemsize == 128
    out, n = C.c_void_p(), C.c_uint64()
    _chk(_L().rth_build_bvh(buf.ctypes.data_as(C.c_void_p), len(buf), C.byref(out), C.byref(n)))
    raw = C.string_at(out.value, n.value * 48)
    _L().rth_free(out)
    return raw


def set_camera(cs, position, orientation, width, height, fov=45.0):
    """The per-frame uniforms of main.cpp:109-123."""
    p = (C.c_float * 3)(*[float(v) for v in position])
    o = (C.c_float * 3)(*[float(v) for v in orientation])
    _chk(_L().rth_set_camera(cs._h, p, o, float(width), float(height), float(fov)))


def orbit_camera(frame):
    """Renderer::UpdateCamera (Renderer.cpp:187-193)."""
    p, o = (C.c_float * 3)(), (C.c_float * 3)()
    _L().rth_orbit_camera(int(frame), p, o)
    return tuple(p), tuple(o)


def frame_rgba8(cs):
    """Renderer::NextFrame's 8-bit conversion (Renderer.cpp:95-124): [H, W, 4] uint8, top row first."""
    out = np.empty((cs.height, cs.width, 4), np.uint8)
    _chk(_L().rth_frame_rgba8(cs._h, cs.width, cs.height, out.ctypes.data_as(C.c_void_p)))
    return out


def decode_image(path):
    """A texture file (PNG or JPEG) as the loader uploads it: [H, W, 4] uint8 (Texture.cpp:5-34)."""
    p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
    _chk(_L().rth_decode_image(str(path).encode(), C.byref(p), C.byref(w), C.byref(h)))
    a = np.frombuffer(C.string_at(p.value, w.value * h.value * 4), np.uint8).reshape(h.value, w.value, 4).copy()
    _L().rth_free(p)
    return a


decode_png = decode_image


def partition(width, height, spp, rank, world, band_ranks=1):
    """rth_partition: the C++ frame driver's shard of `rank` (the same index math as distributed.partition)."""
    out = (C.c_uint32 * 6)()
    _chk(_L().rth_partition(int(width), int(height), int(spp), int(rank), int(world), int(band_ranks), out))
    return tuple(int(v) for v in out)


class MultiGPU:
    """The C++ multi-GPU frame driver (rt_host::MultiGPU, include/rt_host.h rth_multi_*): one process, one context and one
    host thread per entry of `devices`, one RCCL reduce of the colour image onto context 0.  Repeated device ordinals put
    several contexts on one GPU (merged with rt_add_image) — how a single-GPU box rehearses the partition."""

    def __init__(self, devices):
        self._m = C.c_void_p()
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        _chk(_L().rth_multi_create(C.byref(self._m), arr, len(devices)))
        self.width = self.height = 0

    def Delete(self):
        if self._m:
            _L().rth_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.Delete()
        except Exception:
            pass

    def __len__(self):
        return _L().rth_multi_size(self._m)

    def context(self, i):
        """A borrowed ComputeShader view of context i (context 0 holds the merged frame)."""
        from .core import ComputeShader
        cs = ComputeShader.__new__(ComputeShader)
        cs._L = _lib.load()
        cs._h = C.c_void_p(_L().rth_multi_ctx(self._m, int(i)))
        cs.width, cs.height = self.width, self.height
        cs.Delete = lambda: None          # owned by the MultiGPU object
        return cs

    def LoadScene(self, scene_index, models_root):
        _chk(_L().rth_multi_load_scene(self._m, str(models_root).encode(), int(scene_index)))

    def setInt(self, key, value):
        _chk(_L().rth_multi_set_int(self._m, key.encode(), int(value)))

    def resize(self, width, height):
        _chk(_L().rth_multi_resize(self._m, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def set_camera(self, position, orientation, width, height, fov=45.0):
        p = (C.c_float * 3)(*[float(v) for v in position])
        o = (C.c_float * 3)(*[float(v) for v in orientation])
        _chk(_L().rth_multi_set_camera(self._m, p, o, float(width), float(height), float(fov)))

    def render_frame(self, spp, band_ranks=1, merge_aovs=False):
        _chk(_L().rth_multi_render_frame(self._m, int(spp), int(band_ranks), 1 if merge_aovs else 0))

    def used_rccl(self):
        return bool(_L().rth_multi_used_rccl(self._m))
