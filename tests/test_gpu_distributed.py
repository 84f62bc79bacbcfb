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


def _single(scene_id, spp=SPP, region=None, spp_range=None):
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    cs = ComputeShader(device=0)
    S.Scene.LoadScene(cs, scene_id, MODELS)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", BOUNCES)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    if region is None:
        cs.dispatch()
    else:
        cs.dispatch_region(*region, *spp_range)
    imgs = [cs.get_image(k) for k in range(4)]
    cs.Delete()
    return imgs


@pytest.mark.parametrize("scene_id", [4, 3])
def test_cpp_driver_with_one_device_is_the_single_gpu_frame(scene_id):
    """rt_host::MultiGPU with one device: the same frame as rt_render_frame, bit for bit, and RCCL is never touched."""
    from raytracing_amd import scene as S
    from tests.util import same_floats
    want = _single(scene_id)
    m = S.MultiGPU([0])
    assert len(m) == 1
    m.LoadScene(scene_id, MODELS)
    m.setInt("maxBounces", BOUNCES)
    m.resize(W, H)
    m.set_camera(S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    for _ in range(2):
        m.render_frame(SPP)
    assert not m.used_rccl()
    cs0 = m.context(0)
    for k in range(4):
        assert same_floats(cs0.get_image(k), want[k]).all(), k
    m.Delete()


@pytest.mark.parametrize("scene_id,n,band_ranks", [(4, 2, 2), (4, 2, 1), (3, 4, 2), (4, 3, 1)])
def test_cpp_driver_merges_contexts_that_share_the_gpu(scene_id, n, band_ranks):
    """The C++ driver's whole frame path with several contexts (host thread each, textures and buffers replicated) on the
    box's one GPU: shards rendered with rt_render_region and merged with rt_add_image (across GPUs the same sum is the RCCL
    reduce).  Row bands alone reproduce the single-context frame bit for bit; with sample ranges the colour equals the
    float32 sum of the shards in context order, and the AOV images, owned by the shard that holds sample 0, are exact."""
    from raytracing_amd import scene as S
    from tests.util import same_floats
    m = S.MultiGPU([0] * n)
    m.LoadScene(scene_id, MODELS)
    m.setInt("maxBounces", BOUNCES)
    m.resize(W, H)
    m.set_camera(S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    for _ in range(2):                            # twice: the second frame's clear must not see the first frame's sums
        m.render_frame(SPP, band_ranks, merge_aovs=True)
    assert not m.used_rccl()                      # one distinct GPU: no collective
    got = [m.context(0).get_image(k) for k in range(4)]
    m.Delete()
    full = _single(scene_id)
    for k in (1, 2, 3):
        assert same_floats(got[k], full[k]).all(), k
    if band_ranks == n:
        assert same_floats(got[0], full[0]).all()
    else:
        want = None
        for rank in range(n):
            x, y, w, h, r0, r1 = S.partition(W, H, SPP, rank, n, band_ranks)
            img = _single(scene_id, region=(x, y, w, h), spp_range=(r0, r1))[0]
            want = img if want is None else (want + img).astype(np.float32)
        assert same_floats(got[0], want).all()
        assert np.allclose(got[0][..., :3], full[0][..., :3], rtol=1e-5, atol=1e-7)
    assert (got[0][..., 3] == 1.0).all()
