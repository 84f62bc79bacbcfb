#!/usr/bin/env python3
"""Generates tests/golden/vectors/*.npz — small known-answer vectors of the hot path.

The reference (bspafford/Raytracing) ships no tests, fixtures or golden images and cannot be built or run in this
image (DESIGN.md "Oracle"), so these vectors are NOT outputs of the reference: they are outputs of the CPU oracle
(oracle/, math_mode 1 = portable transcendentals) on the reference's own bundled scenes, frozen so that
  * a change of the oracle shows up as a diff against committed data (tests/test_golden.py, CPU), and
  * the HIP path can be checked on the GPU box without the oracle in the loop (tests/test_gpu_golden.py).
Parity with the reference itself stays "unpinned" in the sense of DESIGN.md.

Per scene (default camera main.cpp:239, 48 x 40 pixels — ragged against the 8 x 8 slot tiles): the four output
images at 4 spp / 6 bounces, primary-hit triangle ids, ray counters, and digests of the six SSBO images
(BVH node bytes with the uninitialised pad bytes 12..15 masked).
    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

MODELS = os.path.join(ROOT, "tests", "golden", "models")
OUT = os.path.join(ROOT, "tests", "golden", "vectors")
CASES = {2: (48, 40, 4, 6), 3: (48, 40, 4, 6), 4: (48, 40, 4, 6), 7: (48, 40, 4, 6)}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def masked_nodes(nodes):
    raw = np.frombuffer(np.ascontiguousarray(nodes).tobytes(), np.uint8).reshape(-1, 48).copy()
    raw[:, 12:16] = 0
    return raw


def scene_digests(sc):
    return {"vertices": digest(sc.vertices), "matrices": digest(sc.matrices), "materials": digest(sc.materials),
            "nodes_masked": digest(masked_nodes(sc.nodes)), "triangles": digest(sc.triangles), "lights": digest(sc.lights)}


def main():
    ref.build()
    os.makedirs(OUT, exist_ok=True)
    for scene_id, (W, H, spp, bounces) in CASES.items():
        sc = ref.load_scene(MODELS, scene_id)
        out = ref.render(sc, ref.make_params(sc, W, H, spp, bounces, math_mode=1))
        dg = scene_digests(sc)
        st = out["stats"]
        np.savez_compressed(
            os.path.join(OUT, "scene%d.npz" % scene_id),
            config=np.array([scene_id, W, H, spp, bounces], np.int32),
            color=out["color"], albedo=out["albedo"], normal=out["normal"], depth=out["depth"],
            prim_tri=out["prim_tri"], prim_hit=out["prim_hit"],
            rays=np.array([st["rays_primary"], st["rays_bounce"], st["rays_shadow"]], np.uint64),
            counts=np.array([len(sc.vertices), len(sc.triangles), len(sc.nodes), len(sc.materials), len(sc.lights)], np.int64),
            digest_names=np.array(sorted(dg)), digests=np.array([dg[k] for k in sorted(dg)]))
        print("scene %d: %dx%d spp %d bounces %d, rays %s" % (scene_id, W, H, spp, bounces, st))


if __name__ == "__main__":
    main()
