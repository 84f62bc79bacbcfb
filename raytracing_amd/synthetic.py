"""Procedural GLTF scenes for the large-triangle-count configuration (BASELINE.json configs[3]): a height field of
n x n quads (2 n^2 triangles, (n+1)^2 <= 2^24 vertices because the reference passes indices through float, Q13),
identity node transform, one mesh, one diffuse material, float32 heights from numpy.random.default_rng(1234).
The file pair (.gltf + .bin) goes through the same loader as the bundled models."""
import json
import os

import numpy as np


def heightfield(n, extent=12.0, seed=1234):
    rng = np.random.default_rng(seed)
    xs = np.linspace(-extent, extent, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    ph = rng.random(8, dtype=np.float32) * np.float32(6.28)
    Y = np.zeros_like(X)
    for k in range(4):
        f = np.float32(0.35 * (2 ** k))
        Y += np.float32(1.2 / (2 ** k)) * np.sin(X * f + ph[2 * k]) * np.cos(Z * f + ph[2 * k + 1])
    Y += (rng.random(X.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.5 * extent / n * 2)
    Y += np.float32(1.0)
    P = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    # normals from central differences of the height grid
    dYdx = np.gradient(Y, xs, axis=0).astype(np.float32)
    dYdz = np.gradient(Y, xs, axis=1).astype(np.float32)
    N = np.stack([-dYdx, np.ones_like(Y), -dYdz], -1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    N = N.reshape(-1, 3).astype(np.float32)
    UV = np.stack([(X + extent) / (2 * extent), (Z + extent) / (2 * extent)], -1).reshape(-1, 2).astype(np.float32)
    T = np.zeros((len(P), 4), np.float32)
    T[:, 0] = 1.0
    T[:, 3] = 1.0
    i = np.arange(n, dtype=np.uint32)
    I, J = np.meshgrid(i, i, indexing="ij")
    a = (I * (n + 1) + J).ravel()
    b = a + (n + 1)
    c = a + 1
    d = b + 1
    idx = np.stack([a, c, b, b, c, d], -1).reshape(-1).astype(np.uint32)   # counter-clockwise seen from +y
    return P, N, UV, T, idx


def write_gltf(path, n, extent=12.0, seed=1234, base_color=(0.7, 0.7, 0.72, 1.0), metallic=0.0, roughness=0.6):
    """Writes <path>.gltf and <path>.bin; returns (n_triangles, n_vertices)."""
    assert (n + 1) ** 2 <= 1 << 24, "vertex count must stay <= 2^24 (indices pass through float, model.cpp:575)"
    P, N, UV, T, idx = heightfield(n, extent, seed)
    parts = [P.tobytes(), N.tobytes(), UV.tobytes(), T.tobytes(), idx.tobytes()]
    offs, o = [], 0
    for p in parts:
        offs.append(o)
        o += len(p)
    base = os.path.basename(path)
    with open(path + ".bin", "wb") as f:
        for p in parts:
            f.write(p)
    nv = len(P)
    doc = {
        "asset": {"version": "2.0", "generator": "raytracing_amd.synthetic"},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0, "name": "heightfield"}],
        "materials": [{"name": "ground", "pbrMetallicRoughness": {"baseColorFactor": list(base_color),
                                                                   "metallicFactor": metallic, "roughnessFactor": roughness}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2, "TANGENT": 3}, "indices": 4, "material": 0}]}],
        "accessors": [
            {"bufferView": 0, "componentType": 5126, "count": nv, "type": "VEC3",
             "min": [float(v) for v in P.min(0)], "max": [float(v) for v in P.max(0)]},
            {"bufferView": 1, "componentType": 5126, "count": nv, "type": "VEC3"},
            {"bufferView": 2, "componentType": 5126, "count": nv, "type": "VEC2"},
            {"bufferView": 3, "componentType": 5126, "count": nv, "type": "VEC4"},
            {"bufferView": 4, "componentType": 5125, "count": len(idx), "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": offs[k], "byteLength": len(parts[k])} for k in range(5)],
        "buffers": [{"byteLength": o, "uri": base + ".bin"}],
    }
    with open(path + ".gltf", "w") as f:
        json.dump(doc, f)
    return len(idx) // 3, nv


SYNTHETIC_LIGHTS = [((0.0, 6.75254, 0.0), (1.0, -1.0, 1.0), 1, 100.0)]   # scene 4's sun (Scene.cpp:42)
