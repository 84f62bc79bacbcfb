#!/usr/bin/env python3
"""Headline benchmark: Mrays/s at 1920x1080, 8 bounces (BASELINE.json `metric`), configs[1]:
bundled GLTF scene 4 (diffuse + metal GGX), 1024 spp, on N MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one complete frame (all samples, all bounces) of the hot path with the scene resident in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU): the frame's samples are split into N
contiguous ranges (scene + BVH replicated), every rank traces its range over the whole image and the HDR
colour buffers are summed onto rank 0 with one RCCL reduce — the only collective (strong scaling).
Before the warm-up every rank renders --prepare-frames frames (default 8): the first allocates the path buffers, the
next four are the library's per-scene trial of its traversal kernels, up to three more its trial of the pass shape
(scene preparation, reported in scene_load_s).
Rank 0 prints ONE JSON line.  One ray = one lineCast of shader.comp (camera/bounce segment or shadow ray)
actually traced by the kernels (device counters).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MODELS = os.path.join(ROOT, "tests", "golden", "models")

# BASELINE.json configs (SURVEY.md 8d gives the concrete inputs).  scene 0 = synthetic height field.
WORKLOADS = {
    "c1": dict(scene=3, width=512, height=512, spp=64, bounces=8, grid=0, desc="configs[0]: scene 3 (box + roughSphere, 330 tris, point light, textured)"),
    "c2": dict(scene=4, width=1920, height=1080, spp=1024, bounces=8, grid=0, desc="configs[1]: scene 4 (scene4.gltf, 4802 tris, sun light)"),
    "c3": dict(scene=7, width=1920, height=1080, spp=256, bounces=16, grid=0, desc="configs[2]: scene 7 (RayTracing.gltf, 44722 tris, rough glass, sun light; spp not given by BASELINE, 256 used)"),
    "c5": dict(scene=7, width=3840, height=2160, spp=1024, bounces=8, grid=0, desc="configs[4]: scene 7 orbit (Renderer.cpp:187-193), one 4K frame per step, frame index = step"),
    "c4": dict(scene=0, width=1920, height=1080, spp=256, bounces=8, grid=3162, desc="configs[3]: synthetic height field, 2*grid^2 triangles, sun light"),
}

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md, L2 (per XCD): ~34.5 TB/s aggregate
S_NODE, S_TRI = 64, 48         # bytes per compact node / triangle record
CU_CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: max clock
VMEM_CYCLES_PER_WAVE_LOAD = 17.0  # floor per wave-level load instruction, whatever its width or the number of active lanes (same source)


KERNEL_CHOICE = {0: "closest hit: k_extend, shadow: k_shadow_sun / k_shadow (batch kernels)",
                 1: "closest hit of bounces >= 1: k_extend_refill, shadow: k_shadow_sun / k_shadow",
                 2: "closest hit: k_extend, shadow: k_shadow_refill",
                 3: "closest hit of bounces >= 1: k_extend_refill, shadow: k_shadow_refill",
                 -2: "reference-order walk (k_extend<EXACT>, k_shadow<EXACT>)"}
LLVMPIPE = ("unavailable: the image has libGL/swrast_dri only - no EGL, OSMesa, X server or GLSL compiler to create a headless "
            "context with, llvmpipe lacks GL_ARB_bindless_texture which shader.comp requires (shader.comp:2,364-365), and the "
            "shader text may not travel into this repository; the CPU leg is therefore the oracle port (SURVEY 8d(ii))")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    # scene preparation, before the warm-up: frame 1 allocates the path buffers, frames 2 to 5 are the library's trial of
    # its traversal kernels for this scene (rt_core.h, refill); reported as scene_load_s.kernel_trial_s
    # (+ up to three for the pass shape: one that re-allocates the path buffers, one that is timed, and — if the default shape
    # is kept — one that allocates them back)
    ap.add_argument("--prepare-frames", type=int, default=8)
    # workload = a BASELINE.json config; c2 is the headline (the default), the others are reported for context
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    # development overrides (a run with any of these set is labelled in config and is not the headline)
    ap.add_argument("--scene", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--grid", type=int, default=None, help="c4: quads per side of the synthetic height field (triangles = 2*grid^2)")
    ap.add_argument("--samples-per-pass", type=int, default=0, help="samplesPerPass override (0 = library default)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N ranks on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=INT", help="extra rt_set_int tuning knobs (passStreams, refill, refillIdle, refillBlocksPerCU ... see INTEGRATION.md)")
    ap.add_argument("--dry-run", action="store_true", help="rehearse the launch only: the ranks form the process group (gloo), sum their "
                    "ranks, rank 0 prints a line with n_gpus and nothing touches a GPU (the CPU test of the launcher path)")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    args.overridden = any(getattr(args, k) is not None for k in ("scene", "width", "height", "spp", "bounces", "grid"))
    for k in ("scene", "width", "height", "spp", "bounces", "grid"):
        if getattr(args, k) is None:
            setattr(args, k, wl[k])
    args.desc = wl["desc"]
    return args


def binding_units(args, ex):
    """VALU / texture-addresser busy fractions of the dominant kernel, from the committed counter pass of this workload
    (profiles/pmc_units_<workload>.json, written by tools/pmc_units.py from rocprofv3 --pmc passes): on a cache-resident
    scene these units, not L2 or HBM, are what traversal waits for."""
    path = os.path.join(ROOT, "profiles", "pmc_units_%s.json" % args.workload)
    if args.overridden or not os.path.exists(path):
        return {"valu_busy": None, "ta_busy": None}
    try:
        j = json.load(open(path))
        name = "k_extend_refill" if (ex["kernel_choice"] & 1) and ex["kernel_choice"] >= 0 else "k_extend"
        k = j["kernels"].get(name) or {}
        return {"valu_busy": k.get("valu_busy"), "ta_busy": k.get("ta_busy"), "waves_per_simd": k.get("waves_per_simd"),
                "binding_units_source": "profiles/pmc_units_%s.json (%s; %s)" % (args.workload, name, j.get("source", "rocprofv3 --pmc"))}
    except Exception:
        return {"valu_busy": None, "ta_busy": None}


def cpu_baseline(args, gpu_image):
    """Oracle (CPU restatement of shader.comp, OpenMP) on a bounded sample of the same workload:
    the first s samples r in [0, s) of the frame, s chosen to fill ~args.cpu_seconds."""
    from oracle import ref
    ref.build()
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    os.environ["OMP_NUM_THREADS"] = str(cores)
    if args.scene == 0:
        from oracle import gltf_flatten as G
        from raytracing_amd import synthetic
        flat = G.flatten([args.synthetic_path + ".gltf"])
        flat["lights"], flat["hasSun"] = G.lights_to_gpu(synthetic.SYNTHETIC_LIGHTS)
        sc = ref.Scene(flat)
    else:
        sc = ref.load_scene(MODELS, args.scene)
    W, H = args.width, args.height

    def run(s):
        p = ref.make_params(sc, W, H, args.spp, args.bounces, math_mode=1, spp_range=(0, s))
        t = time.time()
        out = ref.render(sc, p)
        dt = time.time() - t
        st = out["stats"]
        return (st["rays_primary"] + st["rays_bounce"] + st["rays_shadow"]), dt, st

    rays1, dt1, _ = run(1)
    s = max(1, min(args.spp, int(args.cpu_seconds / max(dt1, 1e-3))))
    rays, dt, st = run(s) if s > 1 else (rays1, dt1, None)
    return {"value": round(rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port", "llvmpipe": LLVMPIPE,
            "sample": "scene %d %dx%d, samples r in [0,%d) of %d spp, %d bounces; %.1f s; rays counted as the "
                      "reference casts them (incl. shadow casts the GPU path skips)" % (args.scene, W, H, s, args.spp, args.bounces, dt)}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` does.  This parent has made
    no GPU call (torch is not even imported yet), the ranks are child processes, rank 0's JSON line goes straight to our
    stdout, and we exit with the launcher's status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this host driver
    env["RT_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world):
    seen = 1
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "Mrays/s at 1080p 8-bounce", "value": None, "dry_run": True, "n_gpus": world,
                          "config": {"rccl_ranks": seen, "backend": "gloo" if world > 1 else None}}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line that says n_gpus: 1 for a run that was asked for 8 must be impossible
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a mislabeled number\n" % (args.gpus, world))
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, rank, world)
    import torch
    if args.share_gpu:
        local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    dev = local if world > 1 else 0
    torch.cuda.set_device(dev)

    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    from raytracing_amd.distributed import agree_on_kernels, image_tensor, sample_range
    from raytracing_amd import render_path

    cs = ComputeShader(device=dev, rank=rank, world_size=world)
    if args.scene == 0:
        import tempfile
        from raytracing_amd import synthetic
        args.synthetic_path = os.path.join(tempfile.gettempdir(), "rt_synth_%d_rank%d" % (args.grid, rank))
        ntri, nvert = synthetic.write_gltf(args.synthetic_path, args.grid)
        S.Scene.LoadModels(cs, [args.synthetic_path + ".gltf"], synthetic.SYNTHETIC_LIGHTS)
        args.desc += " (%d triangles, %d vertices)" % (ntri, nvert)
    else:
        S.Scene.LoadScene(cs, args.scene, MODELS)
    load_t = S.Scene.timings()
    W, H = args.width, args.height
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", args.spp)
    cs.setInt("maxBounces", args.bounces)
    if args.samples_per_pass:
        cs.setInt("samplesPerPass", args.samples_per_pass)
    if args.blocks_per_cu:
        cs.setInt("blocksPerCU", args.blocks_per_cu)
    for kv in args.set:
        k, v = kv.split("=")
        cs.setInt(k, int(v))
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    r0, r1 = sample_range(args.spp, rank, world)
    color = image_tensor(cs, 0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    frame_no = [0]

    def step():
        if args.workload == "c5":                    # the camera path of the offline render: frame f of the orbit
            frame_no[0] += 1                         # exported frame f >= 1 is rendered with UpdateCamera(currentFrame = f - 1)
            pos, ori = render_path.frame_camera(frame_no[0])
            S.set_camera(cs, pos, ori, W, H, 45.0)
        cs.dispatch_region(0, 0, W, H, r0, r1)       # returns with the images complete
        if world > 1:
            dist.reduce(color, dst=0, op=dist.ReduceOp.SUM)
            torch.cuda.synchronize()                 # the reduce runs on the process group's stream; the next frame clears `color`
        return cs.stats()

    t_prep = time.perf_counter()
    for _ in range(args.prepare_frames):
        cs.dispatch_region(0, 0, W, H, r0, r1)
    torch.cuda.synchronize()
    if world > 1:
        agree_on_kernels(cs)                          # all ranks run what rank 0's trial kept (one broadcast int)
    load_t = dict(load_t)
    load_t["kernel_trial_s"] = round(time.perf_counter() - t_prep, 3)
    # The timed steps carry no instrumentation: per-kernel HIP events are taken on extra frames after the timed region.
    cs.setInt("timeKernels", 0)
    rccl_ranks = dist.get_world_size() if world > 1 else 1
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    agg = {"rays_primary": 0, "rays_bounce": 0, "rays_shadow": 0, "rays_shadow_skipped": 0,
           "ms_extend": 0.0, "ms_shade": 0.0, "ms_shadow": 0.0, "launches_extend": 0, "launches_shadow": 0}
    last_stats = {}
    for _ in range(args.steps):
        st = step()
        last_stats = st
        for k in agg:
            agg[k] += st[k]
    barrier()
    elapsed = time.perf_counter() - t0
    rays = agg["rays_primary"] + agg["rays_bounce"] + agg["rays_shadow"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda:%d" % dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        r = torch.tensor([float(rays)], dtype=torch.float64, device="cuda:%d" % dev)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays = int(r.item())

    if rank == 0:
        # Algorithmic bytes per closest-hit ray: one instrumented pass (same traversal, counters on) over a few
        # samples of the same frame; per-ray averages x rays of the timed launches / their measured duration.
        cs.setInt("timeKernels", 0)
        cs.setInt("countVisits", 1)
        # (one full pass of the samples the timed frames trace together, library default 64: queue lengths per launch, and
        # with them how often a persistent wave can refill its lanes, are then those of the timed launches)
        cs.dispatch_region(0, 0, W, H, 0, min(args.samples_per_pass or 64, args.spp))
        c = cs.stats()
        cs.setInt("countVisits", 0)
        ext_rays_c = c["rays_primary"] + c["rays_bounce"]
        bytes_per_ray = (S_NODE * c["node_visits"] + S_TRI * c["tri_tests"]) / max(1, ext_rays_c)
        sh_bytes_per_ray = (S_NODE * c["shadow_node_visits"] + S_TRI * c["shadow_tri_tests"]) / max(1, c["rays_shadow"])
        # Launch durations for the roofline.  In the timed region consecutive passes run on several streams (library
        # default passStreams = 4) and their kernels overlap, so a launch's event-to-event time there includes the
        # kernels it shares the GPU with.  The roofline wants the kernel by itself: one more frame of the same
        # workload with passStreams = 1, HIP events on the launch stream (both figures are reported).
        # (one more frame exactly as the timed ones ran, with events on: what a launch lasts beside the other passes' kernels)
        ov = {}
        if not args.no_kernel_timing:
            cs.setInt("timeKernels", 1)
            cs.dispatch_region(0, 0, W, H, r0, r1)
            ov = cs.stats()
        overlapped_ms = ov.get("ms_extend", 0.0) / max(1, ov.get("launches_extend", 1))
        cs.setInt("timeKernels", 1)
        cs.setInt("passStreams", 1)
        cs.setInt("fusePairs", 0)                    # closest-hit and shadow rays in launches of their own, so that each has a duration
        cs.dispatch_region(0, 0, W, H, r0, r1)
        ex = cs.stats()
        cs.setInt("fusePairs", next((int(kv.split("=")[1]) for kv in args.set if kv.startswith("fusePairs=")), 1))
        cs.setInt("passStreams", 0)                  # 0 = back to the library default
        cs.setInt("timeKernels", 0)
        ex_rays = ex["rays_primary"] + ex["rays_bounce"]
        ex_launch = max(1, ex["launches_extend"])
        avg_ms = ex["ms_extend"] / ex_launch
        achieved = (bytes_per_ray * ex_rays / ex_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC counters cannot be collected inside this process (rocprofv3 wraps the
        # command): they come from the committed counter pass of this same command, tools/profile_round.sh ->
        # profiles/traffic_<workload>.json (FETCH_SIZE x 2 + WRITE_SIZE per launch, MI355X_MICROARCH.md HBM section).
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if not args.overridden and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj["kernels"]["k_extend"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/traffic_%s.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % args.workload
            except Exception:
                traffic = None
        alg_launch = bytes_per_ray * ex_rays / ex_launch
        # compact BVH size: one 64-byte record per internal reference node, one 48-byte record per triangle
        import numpy as np
        node_words = np.frombuffer(S.Scene.buffer_view(3), np.uint32).reshape(-1, 12)
        bvh_bytes = int(S_NODE * int((node_words[:, 7] != 1).sum()) + S_TRI * (len(S.Scene.buffer_view(4)) // 128))
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        common = {"kernel": "closest-hit family: k_extend<PRIMARY> (camera rays) + %s (bounce rays)"
                            % ("k_extend_refill" if (ex["kernel_choice"] & 1) and ex["kernel_choice"] >= 0 else "k_extend"),
                  "unit": "GB/s", "traffic": traffic, "traffic_source": traffic_src,
                  "alg_bytes_per_launch": round(alg_launch, 1), "alg_rate_GBs": round(achieved, 2),
                  "compact_bvh_bytes": bvh_bytes,
                  "timing": "exclusive: one frame with passStreams=1 and fusePairs=0 after the timed region, HIP events per launch",
                  "overlapped_avg_launch_ms": round(overlapped_ms, 4),
                  "overlapped_note": "one extra frame as the timed ones run (the timed steps themselves carry no events): passes overlap on 4 streams "
                                     "and the shadow rays of a bounce share a launch with the closest-hit rays of the next (k_trace_pair), so a "
                                     "launch's duration includes co-running work",
                  "exclusive_kernel_ms_per_frame": {k: round(ex[k], 2) for k in ("ms_extend", "ms_extend_primary", "ms_shade", "ms_shadow")},
                  "alg_bytes_per_ray": round(bytes_per_ray, 1), "avg_launch_ms": round(avg_ms, 4),
                  "launches": ex["launches_extend"], "rays_per_launch": round(ex_rays / ex_launch, 1),
                  "nodes_per_ray": round(c["node_visits"] / max(1, ext_rays_c), 2),
                  "tris_per_ray": round(c["tri_tests"] / max(1, ext_rays_c), 2),
                  "shadow_alg_bytes_per_ray": round(sh_bytes_per_ray, 1),
                  "bounce_lane_utilisation": {"node_step": round(c["bounce_node_visits"] / max(1, 64 * c["wave_node_steps"]), 3),
                                              "tri_step": round(c["bounce_tri_tests"] / max(1, 64 * c["wave_tri_steps"]), 3),
                                              "wave_node_steps": c["wave_node_steps"], "wave_tri_steps": c["wave_tri_steps"]},
                  "overlapped_kernel_ms_per_frame": {k: round(ov.get(k, 0.0), 2) for k in ("ms_extend", "ms_shade", "ms_shadow")}}
        common.update(binding_units(args, ex))
        # Which roof serves the node / triangle fetches of this workload.  A compact BVH that fits the L2s and the Infinity
        # Cache (every bundled scene: <= 13 MB) is never fetched from HBM after the first touch (counter traffic is a few per
        # cent of the algorithmic bytes), so its fetch rate is held against the L2 rate of MI355X_MICROARCH.md, and the
        # binding unit, the CU's vector-memory issue path, is reported beside it (issue_floor).  A BVH far larger than the
        # caches (c4: 1.8 GB) is HBM-bound: there the measured HBM bytes per launch (FETCH x 2 + WRITE counter passes of this
        # command, profiles/) over the launch time are held against the HBM peak, and the algorithmic rate is a side figure
        # (caches absorb the top of the tree, so it can exceed what HBM delivers).
        cache_resident = bvh_bytes <= (256 << 20)
        if cache_resident:
            roofline = dict({"bound": "l2", "achieved": round(achieved, 2), "peak": L2_PEAK_GBS,
                             "frac": round(achieved / L2_PEAK_GBS, 4),
                             "algorithmic_rate_over_hbm_peak": round(achieved / HBM_PEAK_GBS, 3),
                             "note": "cache-resident scene: node and triangle records are served by L1/L2 (HBM traffic per launch is `traffic`, a "
                                     "few per cent of the algorithmic bytes), so the roof is the L2 rate (34.5 TB/s, MI355X_MICROARCH.md), not HBM; "
                                     "`algorithmic_rate_over_hbm_peak` is the contract's HBM sanity ratio and is not a fraction of a roof"}, **common)
        else:
            hbm_rate = (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None
            roofline = dict({"bound": "hbm", "achieved": round(hbm_rate, 2) if hbm_rate else None, "peak": HBM_PEAK_GBS,
                             "frac": round(hbm_rate / HBM_PEAK_GBS, 4) if hbm_rate else None,
                             "frac_of_achievable_6300": round(hbm_rate / 6300.0, 4) if hbm_rate else None,
                             "algorithmic_rate_over_hbm_peak": round(achieved / HBM_PEAK_GBS, 3),
                             "note": "BVH far larger than L2 + Infinity Cache: achieved = measured HBM bytes per launch (`traffic`) / average launch "
                                     "time; the algorithmic rate (`alg_rate_GBs`) counts fetches the caches serve and is a side figure"}, **common)
        # The unit that binds traversal of a cache-resident scene: wave-level load instructions through the CU's
        # vector-memory path, >= 17 CU-cycles each however few lanes are active (tools/ubench/gather_rate.hip).  What the
        # bounce rays' closest-hit launches would cost at that floor alone, against what they took: a fraction below 1.
        if c["wave_node_steps"] and c["rays_bounce"]:
            # (two-level nodes: a visit is 7 loads and is counted as two node steps)
            loads_per_node_step = 3.5 if last_stats.get("wide_nodes") else 4
            wave_loads = (loads_per_node_step * c["wave_node_steps"] + 3 * c["wave_tri_steps"]) * (ex["rays_bounce"] / c["rays_bounce"])
            floor_ms = wave_loads * VMEM_CYCLES_PER_WAVE_LOAD / (cus * CU_CLOCK_HZ) * 1e3
            bounce_ms = ex["ms_extend"] - ex["ms_extend_primary"]
            roofline["issue_floor"] = {"floor_ms_per_frame": round(floor_ms, 2), "measured_ms_per_frame": round(bounce_ms, 2),
                                       "frac": round(floor_ms / bounce_ms, 4) if bounce_ms > 0 else None,
                                       "cycles_per_wave_load": VMEM_CYCLES_PER_WAVE_LOAD, "cus": cus, "clock_hz": CU_CLOCK_HZ,
                                       "note": "closest-hit launches of bounces >= 1, exclusive frame: wave-level load instructions (4 per node "
                                               "step, 3 per triangle step; instrumented pass scaled to the frame) x 17 CU-cycles / (CUs x 2.4 GHz)"}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only
            try:
                cpu = cpu_baseline(args, None)
            except Exception as e:  # the baseline is reporting only; never fail the bench on it
                cpu = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "llvmpipe": LLVMPIPE, "sample": "failed: %r" % (e,)}
        headline = (args.workload == "c2" and not args.overridden)
        out = {
            "metric": "Mrays/s at 1080p 8-bounce", "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "bundled scene (tests/golden/models), default camera",
            "config": {"workload": "%s %dx%d, %d spp, %d bounces%s"
                                   % (args.desc, W, H, args.spp, args.bounces, "" if headline else " [NOT THE HEADLINE CONFIG]"),
                       "parallelism": "sample-range x%d, %s reduce of the colour buffer" % (world, "RCCL" if args.backend == "nccl" else args.backend) if world > 1 else "single GPU",
                       "rccl_ranks": rccl_ranks, "backend": args.backend if world > 1 else None,
                       "traversal": "culled/ordered (traversalMode 0)",
                       "traversal_kernels": KERNEL_CHOICE.get(last_stats.get("kernel_choice") if last_stats.get("kernel_choice", 0) < 0 else last_stats.get("kernel_choice") & 3, "?") +
                                            (" on two-level nodes (128-byte records, %d steps deep)" % last_stats.get("wide_depth", 0) if last_stats.get("wide_nodes") else "") +
                                            ("" if last_stats.get("kernel_choice_settled") else " [trial not finished]"),
                       "pass_shape": ("half the pass streams, twice the samples per pass (measured faster for this scene)"
                                      if last_stats.get("kernel_choice", 0) >= 0 and (last_stats.get("kernel_choice", 0) & 4) else "default pass streams and samples per pass")},
            "rays": {"total": rays, "primary": agg["rays_primary"], "bounce": agg["rays_bounce"], "shadow": agg["rays_shadow"],
                     "shadow_skipped": agg["rays_shadow_skipped"], "note": "rank-0 split; total is all ranks" if world > 1 else "per %d steps" % args.steps},
            "scene_load_s": load_t,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    cs.Delete()


if __name__ == "__main__":
    main()
