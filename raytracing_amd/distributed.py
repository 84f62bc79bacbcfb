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
        if color_tensor.is_cuda:
            # The collective runs on the process group's own stream and the core renders on its own streams: wait
            # here, or the next frame's clear of this image would race with the reduce still reading / writing it.
            import torch
            torch.cuda.synchronize(color_tensor.device)
    return color_tensor


def agree_on_kernels(cs, src=0):
    """Every rank of a job runs the traversal kernels rank `src` settled on.  The library picks them per scene from four
    host-clocked trial frames (rt_core.h, refill), so two ranks could decide differently: images would still be identical,
    but shard times, and with them the frame time (the slowest rank's), would not.  The choice (rt_stats.kernel_choice of
    the last render, -1 if that rank's trial has not finished) travels through the existing process group as one integer;
    a settled choice is then pinned on every rank with rt_set_int("kernelChoice").  Returns the choice, or -1."""
    import torch
    import torch.distributed as dist
    st = cs.stats()
    mine = int(st["kernel_choice"]) if st.get("kernel_choice_settled") and st["kernel_choice"] >= 0 else -1
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return mine
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([mine], dtype=torch.int32, device=dev)
    dist.broadcast(t, src=src)
    choice = int(t.item())
    if choice >= 0:
        cs.setInt("kernelChoice", choice)
    return choice


class DeviceImage:
    """Zero-copy torch view of a core image (W*H*4 floats in HBM) through __cuda_array_interface__."""

    def __init__(self, ptr, height, width):
        self.__cuda_array_interface__ = {"shape": (height, width, 4), "typestr": "<f4", "data": (ptr, False), "version": 2}


def image_tensor(cs, image=0):
    import torch
    return torch.as_tensor(DeviceImage(cs.image_device_ptr(image), cs.height, cs.width), device="cuda:%d" % torch.cuda.current_device())


def render_frame_distributed(cs, spp, rank, world, band_ranks=1, dst=0, aovs=False):
    """One frame on `world` GPUs: this rank traces its (row band, sample range) shard with rt_render_region, then the
    colour buffers are summed onto `dst` (and, if asked, the AOV buffers, which are zero on ranks not owning r = 0)."""
    x, y, w, h, r0, r1 = partition(cs.width, cs.height, spp, rank, world, band_ranks)
    if h > 0 and r1 > r0:
        cs.dispatch_region(x, y, w, h, r0, r1)
    else:
        cs.dispatch_region(0, 0, cs.width, min(8, cs.height), 0, 0)   # empty shard: clears the images, traces nothing
    if world > 1:
        for img in ((0, 1, 2, 3) if aovs else (0,)):
            reduce_color(image_tensor(cs, img), dst)
    return (x, y, w, h, r0, r1)
