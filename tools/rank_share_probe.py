#!/usr/bin/env python3
"""What one rank of an N-GPU sample split has to do, timed on one GPU: the sample range [0, spp/N) of the headline frame
(scene 4, 1080p, 1024 spp, 8 bounces) for N = 1, 2, 4, 8 and a few samplesPerPass settings.  Prints ms per frame and
the efficiency t(1) / (N * t(N)) the split can reach before the collective.
usage: python tools/rank_share_probe.py [--scene 4] [--spp 1024] [--bounces 8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", type=int, default=4)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--passes", default="0,32,16")
    args = ap.parse_args()
    import torch
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    cs = ComputeShader(device=0)
    S.Scene.LoadScene(cs, args.scene, os.path.join(ROOT, "tests", "golden", "models"))
    W, H = args.width, args.height
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", args.spp)
    cs.setInt("maxBounces", args.bounces)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    for _ in range(5):                                   # allocation + the library's four-frame kernel trial
        cs.dispatch_region(0, 0, W, H, 0, args.spp)
    base = None
    for n in (1, 2, 4, 8):
        for spass in [int(x) for x in args.passes.split(",")]:
            cs.setInt("samplesPerPass", spass if spass else 64)
            r1 = args.spp // n
            cs.dispatch_region(0, 0, W, H, 0, r1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                cs.dispatch_region(0, 0, W, H, 0, r1)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            if base is None:
                base = ms
            print("N=%d samples [0,%d) samplesPerPass=%d: %.2f ms  efficiency %.3f" % (n, r1, spass if spass else 64, ms, base / (n * ms)), flush=True)


if __name__ == "__main__":
    main()
