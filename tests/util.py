"""Shared helpers for the parity tests: put an oracle-side Scene on the HIP core through the C ABI."""
import numpy as np

from oracle import gltf_flatten as G
from oracle import ref


def setup_core(cs, scene, width, height, spp, bounces, camera=ref.DEFAULT_CAMERA, fov=45.0):
    """Mirrors Scene::LoadGPUData (Scene.cpp:64-76) + Main::Start (main.cpp:243-247) + the per-frame
    uniforms of main.cpp:109-123, all through the drop-in boundary."""
    from raytracing_amd import SSBO
    mats = scene.materials.copy()
    # textures first: handles go into MaterialData (model.cpp:546-559)
    if len(scene.textures):
        handles = [cs.create_texture(t) for t in scene.textures]
        for fld, has in (("baseColorTexture", "hasBaseTexture"), ("normalTexture", "hasNormalTexture"),
                         ("metallicRoughnessTexture", "hasMetallicRoughnessTexture")):
            for m in mats:
                if m[has] == 1:
                    m[fld] = handles[int(m[fld]) - 1]
    SSBO.Bind(cs, scene.vertices, scene.vertices.nbytes, 0)
    SSBO.Bind(cs, scene.matrices, scene.matrices.nbytes, 1)
    SSBO.Bind(cs, mats, mats.nbytes, 2)
    SSBO.Bind(cs, scene.nodes, scene.nodes.nbytes, 3)
    SSBO.Bind(cs, scene.triangles, scene.triangles.nbytes, 4)
    SSBO.Bind(cs, scene.lights, scene.lights.nbytes, 5)
    cs.setInt("numLights", len(scene.lights))
    cs.setInt("hasSun", int(scene.has_sun))
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", bounces)
    set_camera(cs, camera, width, height, fov)
    cs.resize(width, height)


def set_camera(cs, camera, width, height, fov=45.0):
    pos, fwd, right, up = G.camera_basis(*camera)
    cs.setVec3("camPos", pos)
    cs.setVec3("camForward", fwd)
    cs.setVec3("camRight", right)
    cs.setVec3("camUp", up)
    cs.setFloat("aspect", float(np.float32(width) / np.float32(height)))
    cs.setFloat("fov", fov)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_floats(a, b):
    """Bitwise float equality, except that any NaN equals any NaN: x86 SSE produces the default NaN
    0xFFC00000 for invalid operations, gfx950 produces 0x7FC00000 — IEEE-754 leaves the payload open."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def rel_rmse(a, b):
    """Relative RMSE over finite RGB pixels: sqrt(mean((a-b)^2)) / mean(b)."""
    a = np.asarray(a, np.float64)[..., :3]
    b = np.asarray(b, np.float64)[..., :3]
    ok = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    d = (a - b)[ok]
    return float(np.sqrt((d * d).mean()) / max(b[ok].mean(), 1e-12))


def psnr(a, b, peak=1.0):
    a = np.clip(np.nan_to_num(np.asarray(a, np.float64)[..., :3]), 0, peak)
    b = np.clip(np.nan_to_num(np.asarray(b, np.float64)[..., :3]), 0, peak)
    mse = ((a - b) ** 2).mean()
    return float(99.0 if mse == 0 else 10 * np.log10(peak * peak / mse))
