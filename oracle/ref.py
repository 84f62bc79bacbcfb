"""ctypes binding of oracle/liboracle.so (bvh_ref.cpp + rt_ref.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import gltf_flatten as G

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with g++ (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.rt_ref_tan_half_fov.restype = C.c_float
        _LIB.rt_ref_tan_half_fov.argtypes = [C.c_float]
        _LIB.rt_ref_hash13.restype = C.c_float
        _LIB.rt_ref_hash13.argtypes = [C.c_float] * 3
    return _LIB


class RtRefTexture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class RtRefScene(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint64),
                ("matrices", C.c_void_p), ("n_matrices", C.c_uint64),
                ("materials", C.c_void_p), ("n_materials", C.c_uint64),
                ("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
                ("triangles", C.c_void_p), ("n_triangles", C.c_uint64),
                ("lights", C.c_void_p), ("n_lights", C.c_uint64),
                ("textures", C.c_void_p), ("n_textures", C.c_uint64)]


class RtRefParams(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("numLights", C.c_int32), ("hasSun", C.c_int32),
                ("camPos", C.c_float * 3), ("camForward", C.c_float * 3),
                ("camRight", C.c_float * 3), ("camUp", C.c_float * 3),
                ("aspect", C.c_float), ("fov", C.c_float),
                ("samplesPerPixel", C.c_int32), ("maxBounces", C.c_int32),
                ("spp_begin", C.c_int32), ("spp_end", C.c_int32),
                ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("math_mode", C.c_int32), ("write_aovs", C.c_int32)]


class RtRefStats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_bounce", C.c_uint64), ("rays_shadow", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build_bvh(triangles):
    """Model::buildBVH + convertToGPU on 128-byte Triangle records -> NODE_DT array (slot 3)."""
    assert triangles.dtype.itemsize == 128
    buf = np.ascontiguousarray(triangles)
    out = C.c_void_p()
    n = C.c_uint32()
    rc = lib().bvh_ref_build(buf.ctypes.data_as(C.c_void_p), C.c_uint32(len(buf)), C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError("bvh_ref_build failed rc=%d" % rc)
    raw = C.string_at(out.value, n.value * 48)
    lib().bvh_ref_free(out)
    return np.frombuffer(raw, G.NODE_DT).copy()


def decode_texture(path):
    """stbi_load + glTexImage2D(GL_RGBA, ...) semantics of Texture.cpp:5-34: RGBA8, 1-channel -> R only,
    no vertical flip.  PNG/JPEG decoding via PIL (decoders are outside the hot path)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ("L", "1", "P") and im.mode != "P":
        a = np.asarray(im.convert("L"), np.uint8)
        out = np.zeros(a.shape + (4,), np.uint8)
        out[..., 0] = a
        out[..., 3] = 255
        return out
    if im.mode == "RGBA":
        return np.ascontiguousarray(np.asarray(im, np.uint8))
    a = np.asarray(im.convert("RGB"), np.uint8)
    out = np.full(a.shape[:2] + (4,), 255, np.uint8)
    out[..., :3] = a
    return out


class Scene:
    """Holds the six SSBO images (+ decoded textures) and keeps them alive for the C calls."""

    def __init__(self, flat, nodes=None, textures=None):
        self.vertices = np.ascontiguousarray(flat["vertices"])
        self.matrices = np.ascontiguousarray(flat["matrices"], np.float32)
        self.materials = np.ascontiguousarray(flat["materials"])
        self.triangles = np.ascontiguousarray(flat["triangles"])
        self.lights = np.ascontiguousarray(flat["lights"])
        self.has_sun = bool(flat["hasSun"])
        self.nodes = np.ascontiguousarray(nodes if nodes is not None else build_bvh(self.triangles))
        if textures is None:
            textures = [decode_texture(p) for p in flat.get("textures", [])]
        self.textures = [np.ascontiguousarray(t) for t in textures]
        self._tex_arr = (RtRefTexture * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            self._tex_arr[i].rgba8 = t.ctypes.data
            self._tex_arr[i].width = t.shape[1]
            self._tex_arr[i].height = t.shape[0]

    def c_scene(self):
        s = RtRefScene()
        for name, arr in (("vertices", self.vertices), ("matrices", self.matrices), ("materials", self.materials),
                          ("nodes", self.nodes), ("triangles", self.triangles), ("lights", self.lights)):
            setattr(s, name, arr.ctypes.data)
            setattr(s, "n_" + name, len(arr))
        s.textures = C.cast(self._tex_arr, C.c_void_p).value
        s.n_textures = len(self.textures)
        return s


def load_scene(models_root, scene_index):
    return Scene(G.load_scene(models_root, scene_index))


DEFAULT_CAMERA = ((0.0, 3.5, 14.0), (0.0, 0.0, -1.0))   # main.cpp:239, camera.h:21


def make_params(scene, width, height, spp, bounces, camera=DEFAULT_CAMERA, fov=45.0, math_mode=1,
                spp_range=None, region=None, write_aovs=True):
    pos, fwd, right, up = G.camera_basis(*camera)
    p = RtRefParams()
    p.width, p.height = width, height
    p.numLights, p.hasSun = len(scene.lights), int(scene.has_sun)
    p.camPos[:] = [float(v) for v in pos]
    p.camForward[:] = [float(v) for v in fwd]
    p.camRight[:] = [float(v) for v in right]
    p.camUp[:] = [float(v) for v in up]
    p.aspect = float(np.float32(width) / np.float32(height))   # main.cpp:122
    p.fov = fov
    p.samplesPerPixel, p.maxBounces = spp, bounces
    p.spp_begin, p.spp_end = spp_range if spp_range else (0, spp)
    p.x0, p.y0, p.x1, p.y1 = region if region else (0, 0, width, height)
    p.math_mode = math_mode
    p.write_aovs = int(write_aovs)
    return p


def render(scene, params):
    """Returns dict(color, albedo, normal, depth [H,W,4] float32; prim_tri [H,W] int32; prim_hit [H,W,3]; stats)."""
    W, H = params.width, params.height
    imgs = {k: np.zeros((H, W, 4), np.float32) for k in ("color", "albedo", "normal", "depth")}
    prim = np.full((H, W), -2, np.int32)
    prim_hit = np.zeros((H, W, 3), np.float32)
    st = RtRefStats()
    cs = scene.c_scene()
    fp = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().rt_ref_render(C.byref(cs), C.byref(params), fp(imgs["color"]), fp(imgs["albedo"]), fp(imgs["normal"]),
                             fp(imgs["depth"]), fp(prim), fp(prim_hit), C.byref(st))
    assert rc == 0
    imgs["prim_tri"] = prim
    imgs["prim_hit"] = prim_hit
    imgs["stats"] = st.as_dict()
    return imgs


def linecast(scene, origins, dirs):
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    n = len(o)
    tri = np.zeros(n, np.int32)
    hit = np.zeros((n, 3), np.float32)
    st = RtRefStats()
    cs = scene.c_scene()
    lib().rt_ref_linecast(C.byref(cs), o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_uint64(n),
                          tri.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p), C.byref(st))
    return tri, hit, st.as_dict()
