"""ctypes binding of librt_core.so (include/rt_core.h).  Fails loudly when the library is missing:
there is no CPU or PyTorch fallback for the integrator."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_CORE_LIB: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("RT_CORE_LIB") or os.path.join(_HERE, "librt_core.so")

EXPORTS = ["rt_create", "rt_destroy", "rt_last_error", "rt_upload_buffer", "rt_set_int", "rt_set_float",
           "rt_set_vec2", "rt_set_vec3", "rt_set_vec4", "rt_set_mat4", "rt_resize", "rt_create_texture",
           "rt_render_frame", "rt_render_region", "rt_read_image", "rt_write_image", "rt_read_image_rgb", "rt_write_image_rgb", "rt_read_frame_rgba8", "rt_image_device_ptr",
           "rt_add_image", "rt_get_stats", "rt_read_primary_hits", "rt_trace_rays", "rt_trace_rays_noalloc"]


class RtConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32), ("reserved", C.c_int32)]


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("rays_shadow_skipped", C.c_uint64), ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shadow_node_visits", C.c_uint64), ("shadow_tri_tests", C.c_uint64), ("max_ray_steps", C.c_uint64),
                ("ms_total", C.c_double), ("ms_extend", C.c_double), ("ms_shade", C.c_double), ("ms_shadow", C.c_double),
                ("launches_extend", C.c_uint64), ("launches_shade", C.c_uint64), ("launches_shadow", C.c_uint64),
                ("bounce_node_visits", C.c_uint64), ("bounce_tri_tests", C.c_uint64),
                ("wave_node_steps", C.c_uint64), ("wave_tri_steps", C.c_uint64),
                ("ms_extend_primary", C.c_double), ("launches_extend_primary", C.c_uint64),
                ("kernel_choice", C.c_int32), ("kernel_choice_settled", C.c_int32),
                ("launches_pair", C.c_uint64), ("wide_nodes", C.c_int32), ("wide_depth", C.c_int32),
                ("log_dense_bounces", C.c_int32), ("log_fallbacks", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


NOALLOC_MAX_RAYS = 16384      # RT_NOALLOC_MAX_RAYS of include/rt_core.h

_lib = None


def load():
    """Returns the loaded library; raises if it has not been built (python -m raytracing_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("librt_core.so is missing at %s — build it with `python -m raytracing_amd.build` "
                           "(hipcc, gfx950).  The integrator has no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_float
    sig = {
        "rt_create": (i32, [C.POINTER(vp), C.POINTER(RtConfig)]),
        "rt_destroy": (None, [vp]),
        "rt_last_error": (C.c_char_p, [vp]),
        "rt_upload_buffer": (i32, [vp, u32, vp, u64]),
        "rt_set_int": (i32, [vp, C.c_char_p, i32]),
        "rt_set_float": (i32, [vp, C.c_char_p, f]),
        "rt_set_vec2": (i32, [vp, C.c_char_p, C.POINTER(f)]),
        "rt_set_vec3": (i32, [vp, C.c_char_p, C.POINTER(f)]),
        "rt_set_vec4": (i32, [vp, C.c_char_p, C.POINTER(f)]),
        "rt_set_mat4": (i32, [vp, C.c_char_p, C.POINTER(f)]),
        "rt_resize": (i32, [vp, u32, u32]),
        "rt_create_texture": (i32, [vp, vp, u32, u32, C.POINTER(u64)]),
        "rt_render_frame": (i32, [vp]),
        "rt_render_region": (i32, [vp, u32, u32, u32, u32, u32, u32]),
        "rt_read_image": (i32, [vp, u32, vp]),
        "rt_write_image": (i32, [vp, u32, vp]),
        "rt_read_image_rgb": (i32, [vp, u32, vp]),
        "rt_write_image_rgb": (i32, [vp, u32, vp]),
        "rt_read_frame_rgba8": (i32, [vp, vp, C.c_uint64]),
        "rt_image_device_ptr": (i32, [vp, u32, C.POINTER(vp)]),
        "rt_add_image": (i32, [vp, u32, vp]),
        "rt_get_stats": (i32, [vp, C.POINTER(RtStats)]),
        "rt_read_primary_hits": (i32, [vp, vp, vp]),
        "rt_trace_rays": (i32, [vp, vp, vp, u64, vp, vp]),
        "rt_trace_rays_noalloc": (i32, [vp, vp, vp, u64, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
