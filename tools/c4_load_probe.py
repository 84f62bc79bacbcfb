#!/usr/bin/env python3
"""GPU box: what loading config 4's scene costs, step by step: GLTF flatten, BVH build, the six uploads, and the first render
call's derivation + upload of the compact device layout (timed on a 16x16, 1-sample render)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import ComputeShader, synthetic, scene as S
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 3162
p = os.path.join(tempfile.gettempdir(), "hf%d" % grid)
synthetic.write_gltf(p, grid)
cs = ComputeShader()
out = []
for rnd in range(2):
    t = time.time(); S.Scene.LoadModels(cs, [p + ".gltf"], synthetic.SYNTHETIC_LIGHTS); load = time.time() - t
    cs.resize(16, 16); cs.setInt("samplesPerPixel", 1); cs.setInt("maxBounces", 1)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], 16, 16, 45.0)
    t = time.time(); cs.dispatch(); first = time.time() - t
    t = time.time(); cs.dispatch(); second = time.time() - t
    out.append(dict(S.Scene.timings(), load_total_s=round(load, 3), first_render_s=round(first, 3), second_render_s=round(second, 4)))
print(json.dumps({"grid": grid, "rounds": out}))
