#!/usr/bin/env python3
"""How long is the longest closest-hit ray of a workload?  One counted pass (countVisits = 1, batch kernels) of a bench
workload; prints rt_stats.max_ray_steps (node visits + triangle tests of a single ray) and the averages beside it.
usage: tools/tail_probe.py <workload> [samples]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1]]
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 64
from raytracing_amd import ComputeShader  # noqa: E402
from raytracing_amd import scene as S  # noqa: E402

cs = ComputeShader(device=0)
S.Scene.LoadScene(cs, wl["scene"], bench.MODELS)
cs.resize(wl["width"], wl["height"])
cs.setInt("samplesPerPixel", wl["spp"])
cs.setInt("maxBounces", wl["bounces"])
S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], wl["width"], wl["height"], 45.0)
cs.setInt("countVisits", 1)
cs.dispatch_region(0, 0, wl["width"], wl["height"], 0, samples)
st = cs.stats()
rays = st["rays_primary"] + st["rays_bounce"]
print({"workload": sys.argv[1], "samples": samples, "max_ray_steps": st["max_ray_steps"],
       "mean_steps_per_ray": round((st["node_visits"] + st["tri_tests"]) / max(1, rays), 2), "closest_hit_rays": rays})
cs.Delete()
