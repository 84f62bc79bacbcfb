"""Python host mirror of the reference's GL-side operator API, over the C ABI (ctypes).

Names and argument meaning follow the reference (computeShader.h:11-31, SSBO.h:7-10):
    cs = ComputeShader()                      # computeShader.cpp:7  (the "program" = the HIP core)
    SSBO.Bind(cs, data, size, slot)           # SSBO.cpp:5  (GL has an implicit context; here it is cs)
    cs.setInt("samplesPerPixel", 100) ...     # computeShader.cpp:53-75
    cs.resize(w, h); cs.dispatch(); img = cs.get_image(0)
Unknown uniform names are ignored like GL location -1.
"""
import ctypes as C

import numpy as np

from . import _lib


class RtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("rt_core status %d: %s" % (status, msg))
        self.status = status


class ComputeShader:
    def __init__(self, device=0, rank=0, world_size=1):
        self._L = _lib.load()
        self._h = C.c_void_p()
        cfg = _lib.RtConfig(device, rank, world_size, 0)
        st = self._L.rt_create(C.byref(self._h), C.byref(cfg))
        if st != 0:
            raise RtError(st, (self._L.rt_last_error(None) or b"").decode())
        self.width = self.height = 0

    # -- lifetime ------------------------------------------------------------------------------
    def Delete(self):
        if self._h:
            self._L.rt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Delete()
        except Exception:
            pass

    def Activate(self):   # glUseProgram: nothing to do, one program per context
        pass

    def _chk(self, st):
        if st != 0:
            raise RtError(st, (self._L.rt_last_error(self._h) or b"").decode())

    # -- uniforms ------------------------------------------------------------------------------
    def setInt(self, key, value):
        self._chk(self._L.rt_set_int(self._h, key.encode(), int(value)))

    def setFloat(self, key, value):
        self._chk(self._L.rt_set_float(self._h, key.encode(), float(value)))

    def _vec(self, fn, key, value, n):
        a = (C.c_float * n)(*[float(v) for v in np.asarray(value, np.float32).reshape(-1)[:n]])
        self._chk(fn(self._h, key.encode(), a))

    def setVec2(self, key, value):
        self._vec(self._L.rt_set_vec2, key, value, 2)

    def setVec3(self, key, value):
        self._vec(self._L.rt_set_vec3, key, value, 3)

    def setVec4(self, key, value):
        self._vec(self._L.rt_set_vec4, key, value, 4)

    def setMat4(self, key, value):
        self._vec(self._L.rt_set_mat4, key, value, 16)

    # -- buffers / images ------------------------------------------------------------------------
    def upload_buffer(self, slot, data, size=None):
        buf = np.ascontiguousarray(data)
        nbytes = buf.nbytes if size is None else int(size)
        self._chk(self._L.rt_upload_buffer(self._h, slot, buf.ctypes.data_as(C.c_void_p), nbytes))

    def create_texture(self, rgba8):
        a = np.ascontiguousarray(rgba8, np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        h = C.c_uint64()
        self._chk(self._L.rt_create_texture(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[0], C.byref(h)))
        return h.value

    def resize(self, width, height):
        self._chk(self._L.rt_resize(self._h, int(width), int(height)))
        self.width, self.height = int(width), int(height)

    def dispatch(self):
        """The whole 64x64-tile glDispatchCompute loop + glMemoryBarrier of main.cpp:126-141."""
        self._chk(self._L.rt_render_frame(self._h))

    def dispatch_region(self, x, y, w, h, spp_begin, spp_end):
        self._chk(self._L.rt_render_region(self._h, x, y, w, h, spp_begin, spp_end))

    def get_image(self, image=0):
        """glGetTexImage(GL_RGBA, GL_FLOAT): [H, W, 4] float32, row 0 = bottom."""
        out = np.empty((self.height, self.width, 4), np.float32)
        self._chk(self._L.rt_read_image(self._h, image, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_image(self, image, rgba):
        a = np.ascontiguousarray(rgba, np.float32)
        assert a.shape == (self.height, self.width, 4)
        self._chk(self._L.rt_write_image(self._h, image, a.ctypes.data_as(C.c_void_p)))

    def get_image_rgb(self, image=0):
        """main.cpp:157-170 (glGetTexImage + rgbaTOrgb): [H, W, 3] float32 in OIDN's Float3 layout."""
        out = np.empty((self.height, self.width, 3), np.float32)
        self._chk(self._L.rt_read_image_rgb(self._h, image, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_image_rgb(self, image, rgb):
        """main.cpp:176-184: a denoised Float3 buffer back into the RGBA image, alpha 1."""
        a = np.ascontiguousarray(rgb, np.float32)
        assert a.shape == (self.height, self.width, 3)
        self._chk(self._L.rt_write_image_rgb(self._h, image, a.ctypes.data_as(C.c_void_p)))

    def image_device_ptr(self, image=0):
        p = C.c_void_p()
        self._chk(self._L.rt_image_device_ptr(self._h, image, C.byref(p)))
        return p.value

    def stats(self):
        s = _lib.RtStats()
        self._chk(self._L.rt_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def primary_hits(self):
        tri = np.empty((self.height, self.width), np.int32)
        hit = np.empty((self.height, self.width, 3), np.float32)
        self._chk(self._L.rt_read_primary_hits(self._h, tri.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p)))
        return tri, hit

    def trace_rays(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        tri = np.empty(len(o), np.int32)
        hit = np.empty((len(o), 3), np.float32)
        self._chk(self._L.rt_trace_rays(self._h, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), len(o),
                                        tri.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p)))
        return tri, hit

    def trace_rays_noalloc(self, origins, dirs):
        """rt_trace_rays_noalloc: the same query out of the library's file-scope scratch array, no allocation on the GPU;
        at most _lib.NOALLOC_MAX_RAYS rays, scene already resident."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        assert o.shape == d.shape
        tri = np.empty(len(o), np.int32)
        hit = np.empty((len(o), 3), np.float32)
        self._chk(self._L.rt_trace_rays_noalloc(self._h, o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), len(o),
                                                tri.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p)))
        return tri, hit


class SSBO:
    """SSBO::Bind(data, size, slot) (SSBO.cpp:5-17) — GL's implicit context is the explicit first argument."""

    @staticmethod
    def Bind(cs, data, size, slot):
        cs.upload_buffer(slot, data, size)
