#!/usr/bin/env python3
"""Time one rank's shard of the headline frame (samples [0, spp/N) of 1024) for N = 1, 2, 4, 8 and a few samplesPerPass:
what strong scaling over sample ranges costs per rank (no collective here).  tools/shard_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import ComputeShader
from raytracing_amd import scene as S
models = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models")
W, H, spp = 1920, 1080, 1024
cs = ComputeShader()
S.Scene.LoadScene(cs, 4, models)
cs.resize(W, H); cs.setInt("samplesPerPixel", spp); cs.setInt("maxBounces", 8)
S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
cs.dispatch_region(0, 0, W, H, 0, 64)
t1 = 1.0
for n in (1, 2, 4, 8, 16):
    for sp in (0, 64, 32, 16):
        cs.setInt("samplesPerPass", sp if sp else 64)
        cs.setInt("autoPass", 1 if sp == 0 else 0)
        r1 = spp // n
        cs.dispatch_region(0, 0, W, H, 0, r1)
        t = time.perf_counter()
        for _ in range(2):
            cs.dispatch_region(0, 0, W, H, 0, r1)
        dt = (time.perf_counter() - t) / 2
        if n == 1 and sp == 0:
            t1 = dt
        print("N=%2d samples %4d samplesPerPass %4s: %7.2f ms  (ideal %.2f ms = frame / N: %5.1f %%)" % (n, r1, sp if sp else "auto", dt * 1e3, t1 * 1e3 / n, 100 * t1 / n / dt), flush=True)
