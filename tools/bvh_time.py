#!/usr/bin/env python3
"""Times the product's BVH builder on the synthetic height field of config 4 (host only unless a GPU is present and the
core has registered its sweep kernels).  usage: RT_BVH_TIMING=1 tools/bvh_time.py [grid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import synthetic, scene as S

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
P, N, UV, T, idx = synthetic.heightfield(grid)
tri = idx.reshape(-1, 3)
f32 = np.float32
TRI_DT = np.dtype({"names": ["indices", "p1", "p2", "p3", "centroidLoc", "minBounds", "maxBounds", "meshIndex", "materialIndex"],
                   "formats": [(np.uint32, 3), (f32, 3), (f32, 3), (f32, 3), (f32, 3), (f32, 3), (f32, 3), np.uint32, np.uint32],
                   "offsets": [0, 16, 32, 48, 64, 80, 96, 108, 112], "itemsize": 128})      # model.h:41-70
rec = np.zeros(len(tri), TRI_DT)
a, b, c = P[tri[:, 0]], P[tri[:, 1]], P[tri[:, 2]]
rec["indices"] = tri
rec["p1"], rec["p2"], rec["p3"] = a, b, c
rec["centroidLoc"] = (a + b + c) / np.float32(3.0)
rec["minBounds"] = np.minimum(np.minimum(a, b), c)
rec["maxBounds"] = np.maximum(np.maximum(a, b), c)
t = time.time()
nodes = S.build_bvh(rec)
print("grid %d: %d triangles, %d nodes, %.3f s" % (grid, len(rec), len(nodes), time.time() - t))
