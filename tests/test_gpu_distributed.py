"""GPU: the multi-GPU frame driver end to end on the box's one GPU — two ranks (two processes, each with its own core
context on cuda:0) render their shards with rt_render_region and merge them with the product's one collective, here
over gloo because RCCL refuses two ranks on one device.  Row-band split must reproduce the single-process frame bit
for bit, sample split within float re-association; the AOV images are exact in both."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models")
W, H, SPP, BOUNCES = 96, 64, 6, 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, band_ranks, out_path):
    import torch                                  # before the core: see render_path.py
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    from raytracing_amd.distributed import render_frame_distributed
    cs = ComputeShader(device=0, rank=rank, world_size=world)
    S.Scene.LoadScene(cs, 4, MODELS)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", SPP)
    cs.setInt("maxBounces", BOUNCES)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    for _ in range(2):                            # twice: the second frame's clear must not race with the first reduce
        shard = render_frame_distributed(cs, SPP, rank, world, band_ranks, aovs=True)
    if rank == 0:
        np.savez(out_path, shard=np.array(shard), **{"img%d" % k: cs.get_image(k) for k in range(4)})
    dist.barrier()
    dist.destroy_process_group()
    cs.Delete()


@pytest.mark.parametrize("band_ranks", [2, 1])
def test_two_ranks_on_one_gpu_merge_to_the_single_process_frame(tmp_path, band_ranks):
    import torch.multiprocessing as mp
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(2, _free_port(), band_ranks, out), nprocs=2, join=True)
    got = np.load(out)

    cs = ComputeShader(device=0)
    S.Scene.LoadScene(cs, 4, MODELS)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", SPP)
    cs.setInt("maxBounces", BOUNCES)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    cs.dispatch()
    full = [cs.get_image(k) for k in range(4)]
    cs.Delete()

    # all four channels: alpha is 1 in the merged frame too (only the rank whose range holds sample 0 stores it)
    color, want = got["img0"], full[0]
    assert (color[..., 3] == 1.0).all()
    if band_ranks == 2:
        assert np.array_equal(color.view(np.uint32), want.view(np.uint32))
    else:
        assert np.allclose(color, want, rtol=1e-5, atol=1e-6)
    for k in (1, 2, 3):      # AOVs come from sample 0 alone: exact under either split
        assert np.array_equal(got["img%d" % k].view(np.uint32), full[k].view(np.uint32)), k
