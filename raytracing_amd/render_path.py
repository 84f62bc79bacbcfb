#!/usr/bin/env python3
"""Config 5 of BASELINE.json: the reference's offline render (Renderer::Start / NextFrame / UpdateCamera,
Renderer.cpp:19-193) — a 120-frame orbit of the scene — on N MI355X, headless.

    python -m raytracing_amd.render_path --scene 7 --width 3840 --height 2160 --spp 1024 --bounces 8 --frames 120 --out Renders/
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m raytracing_amd.render_path ...

Frame f is rendered with the camera the reference would have at that frame: frame 0 with the camera the user left
(the default camera here, main.cpp:239), frame f >= 1 with UpdateCamera's orbit for currentFrame = f - 1: NextFrame
encodes frame f - 1, calls UpdateCamera and only then increments currentFrame (Renderer.cpp:151,164,187-193), and
main.cpp sets the uniforms and dispatches before it calls NextFrame (main.cpp:111,138,198).  Two ways to use N GPUs:
  --split samples (default)  every frame is split into N sample ranges (x row bands with --band-ranks) and the HDR
                             colour buffers are summed onto rank 0 with one RCCL reduce (the only collective);
  --split frames             rank r renders frames r, r + N, ... on its own: no collective at all.
FFmpeg is not in this image, so instead of an H.264 stream the 8-bit frames (clamp, truncate, vertical flip exactly as
Renderer.cpp:103-124) are written as binary PPM files Scene-<i>_spp-<s>_bounces-<b>/frame_%04d.ppm; MP4 export stays
a host-side consumer of those frames.
"""
import argparse
import json
import os
import sys
import time


def frame_camera(f):
    """Camera of exported frame f: the user's camera for frame 0, UpdateCamera with currentFrame = f - 1 after it."""
    from . import scene as S
    return S.DEFAULT_CAMERA if f == 0 else S.orbit_camera(f - 1)


def write_ppm(path, rgba8):
    h, w = rgba8.shape[:2]
    with open(path, "wb") as fh:
        fh.write(b"P6\n%d %d\n255\n" % (w, h))
        fh.write(rgba8[..., :3].tobytes())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", type=int, default=7)
    ap.add_argument("--models", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "models"))
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--frames", type=int, default=120)           # Renderer.h:32 totalFrames
    ap.add_argument("--first-frame", type=int, default=0)
    ap.add_argument("--split", choices=["samples", "frames"], default="samples")
    ap.add_argument("--band-ranks", type=int, default=1, help="--split samples: row bands x (world / bands) sample ranges")
    ap.add_argument("--out", default=None, help="directory for the PPM frames (omit: render only)")
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    # torch is only the collective's plumbing: a single process never touches it.  With several ranks it is
    # initialised BEFORE the core (two HIP runtimes in one process: the one that comes second finds no GPU).
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    from . import ComputeShader
    from . import scene as S
    from .distributed import agree_on_kernels, render_frame_distributed

    cs = ComputeShader(device=local, rank=rank, world_size=world)
    S.Scene.LoadScene(cs, args.scene, args.models)
    cs.resize(args.width, args.height)
    cs.setInt("samplesPerPixel", args.spp)
    cs.setInt("maxBounces", args.bounces)
    out_dir = None
    if args.out:
        out_dir = os.path.join(args.out, "Scene-%d_spp-%d_bounces-%d" % (args.scene, args.spp, args.bounces))   # Renderer.cpp:30
        os.makedirs(out_dir, exist_ok=True)

    frames = range(args.first_frame, args.first_frame + args.frames)
    rays = 0
    agreed = False
    from concurrent.futures import ThreadPoolExecutor
    pool, pending = ThreadPoolExecutor(max_workers=1), []
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
    t0 = time.perf_counter()
    for f in frames:
        if args.split == "frames" and f % world != rank:
            continue
        pos, ori = frame_camera(f)
        S.set_camera(cs, pos, ori, args.width, args.height, 45.0)
        if args.split == "frames":
            cs.dispatch()
            writer = True
        else:
            render_frame_distributed(cs, args.spp, rank, world, args.band_ranks)
            writer = rank == 0
        st = cs.stats()
        if world > 1 and args.split == "samples" and not agreed and f - args.first_frame >= 4:
            agreed = agree_on_kernels(cs) >= 0        # once rank 0's per-scene trial has finished, every rank runs its choice
        rays += st["rays_primary"] + st["rays_bounce"] + st["rays_shadow"]
        if out_dir and writer:
            # the 8-bit frame is converted on the GPU and read back here; the file is written by a worker thread while
            # the next frame renders (the render call releases the GIL); at most two frames are held
            if len(pending) >= 2:
                pending.pop(0).result()
            pending.append(pool.submit(write_ppm, os.path.join(out_dir, "frame_%04d.ppm" % f), S.frame_rgba8(cs)))
    for fut in pending:                # frames still being written count
        fut.result()
    pool.shutdown()
    if world > 1:                      # (dispatch returns with the images complete)
        torch.cuda.synchronize()
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        r = torch.tensor([float(rays)], dtype=torch.float64, device="cuda:%d" % local)
        dist.all_reduce(r)
        rays = int(r.item())
    if rank == 0:
        print(json.dumps({"scene": args.scene, "width": args.width, "height": args.height, "spp": args.spp, "bounces": args.bounces,
                          "frames": args.frames, "n_gpus": world, "split": args.split, "seconds": round(dt, 3),
                          "frames_per_s": round(args.frames / dt, 4), "Mrays_per_s": round(rays / dt / 1e6, 1), "out": out_dir}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    cs.Delete()


if __name__ == "__main__":
    main()
