"""Multi-GPU partition of one frame.  Pure index math + one RCCL collective.

Scene and BVH are replicated on every GPU.  A frame is split by contiguous sample ranges [r0, r1) of
samplesPerPixel (every rank keeps spp unchanged, so the stratified grid, the 1/spp strength and the
r*bounce hash seeds are identical to the single-GPU run) and/or by row bands of 8-pixel tile rows.
Colour is a plain sum over samples (shader.comp:312,332,521): ranks' partial images are summed with one
reduce (disjoint bands contribute zeros, so the same collective covers both splits).  AOV images come from
sample 0 only and are written by the ranks whose range contains r = 0.
"""


def sample_range(spp, rank, world):
    """Contiguous, balanced split of range(spp): ranks < spp % world get one extra sample."""
    base, extra = divmod(int(spp), int(world))
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def row_band(height, rank, world, align=8):
    """Balanced split of the image rows into bands whose boundaries are multiples of `align`
    (the wavefront tile height), so a band maps to whole 8x8 ray tiles."""
    tiles = (int(height) + align - 1) // align
    base, extra = divmod(tiles, int(world))
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * align, height), min(t1 * align, height)


def partition(width, height, spp, rank, world, band_ranks=1):
    """Tile x sample-range partition: `band_ranks` row bands x (world // band_ranks) sample ranges.
    Returns (x, y, w, h, r0, r1) for rt_render_region."""
    assert world % band_ranks == 0
    sample_ranks = world // band_ranks
    b, s = divmod(rank, sample_ranks)
    y0, y1 = row_band(height, b, band_ranks)
    r0, r1 = sample_range(spp, s, sample_ranks)
    return 0, y0, width, y1 - y0, r0, r1


def reduce_color(color_tensor, dst=0):
    """The only collective: sum the HDR accumulation buffers onto `dst` (RCCL over xGMI via torch.distributed)."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(color_tensor, dst=dst, op=dist.ReduceOp.SUM)
    return color_tensor
