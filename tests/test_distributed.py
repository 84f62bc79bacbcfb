"""CPU: the multi-GPU partition (raytracing_amd/distributed.py) — index math, and a world_size-2 gloo run in which
each rank renders its shard (with the CPU oracle standing in for the GPU core) and the colour buffers are merged by
the product's one collective.  Tile split must be bit-identical to the single-rank frame, sample split within float
re-association."""
import os
import socket

import numpy as np
import pytest

from raytracing_amd import distributed as D

MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models")


def test_sample_range_covers_exactly():
    for spp in (1, 7, 64, 1024, 1000):
        for world in (1, 2, 3, 4, 8):
            rs = [D.sample_range(spp, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_row_bands_cover_exactly_and_align():
    for h in (8, 54, 600, 1080, 2160, 13):
        for world in (1, 2, 4, 8):
            bs = [D.row_band(h, r, world) for r in range(world)]
            assert bs[0][0] == 0 and bs[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(bs, bs[1:]))
            assert all(a % 8 == 0 for a, b in bs if b > a)


def test_partition_is_a_partition_of_pixel_samples():
    W, H, spp = 64, 40, 10
    for world, bands in ((2, 1), (2, 2), (4, 2), (8, 2), (8, 8), (8, 1)):
        cover = np.zeros((H, spp), np.int32)
        for r in range(world):
            x, y, w, h, r0, r1 = D.partition(W, H, spp, r, world, bands)
            assert x == 0 and w == W
            cover[y:y + h, r0:r1] += 1
        assert (cover == 1).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, band_ranks, out_path):
    import torch
    import torch.distributed as dist
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from oracle import ref
    sc = ref.load_scene(MODELS, 4)
    W, H, spp, b = 48, 32, 4, 5
    x, y, w, h, r0, r1 = D.partition(W, H, spp, rank, world, band_ranks)
    p = ref.make_params(sc, W, H, spp, b, spp_range=(r0, r1), region=(x, y, x + w, y + h), write_aovs=(r0 == 0))
    img = ref.render(sc, p)
    color = torch.from_numpy(img["color"].copy())
    color[..., 3] = 0 if rank else color[..., 3]
    albedo = torch.from_numpy(img["albedo"].copy())
    D.reduce_color(color, dst=0)          # the product's collective
    D.reduce_color(albedo, dst=0)
    if rank == 0:
        np.savez(out_path, color=color.numpy(), albedo=albedo.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("band_ranks", [2, 1])
def test_world_size_2_gloo_merge(tmp_path, band_ranks, oracle_lib):
    import torch.multiprocessing as mp
    out = str(tmp_path / "merged.npz")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, band_ranks, out), nprocs=2, join=True)
    got = np.load(out)
    sc = oracle_lib.load_scene(MODELS, 4)
    full = oracle_lib.render(sc, oracle_lib.make_params(sc, 48, 32, 4, 5))
    if band_ranks == 2:     # tile split only: bit-identical
        assert np.array_equal(got["color"][..., :3].view(np.uint32), full["color"][..., :3].view(np.uint32))
    else:                   # sample split: float sums re-associate
        assert np.allclose(got["color"][..., :3], full["color"][..., :3], rtol=1e-5, atol=1e-6)
    assert np.array_equal(got["albedo"][..., :3], full["albedo"][..., :3])
