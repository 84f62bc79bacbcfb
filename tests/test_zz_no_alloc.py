"""GPU: rt_trace_rays_noalloc — the closest-hit query out of the library's file-scope scratch array — against
rt_trace_rays (its allocating counterpart, itself checked against the oracle in test_gpu_parity.py), bit for bit, through
the C ABI: two calls in a row, the second with a smaller input (the array keeps the first call's contents), then the
limits.  Collected last on purpose (the round's end runs pytest -x)."""
import ctypes

import numpy as np
import pytest

from tests.util import bits, setup_core

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    """Rays from around scene 7 towards its middle, a few of them degenerate (zero direction components, misses)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-9.0, 9.0, (n, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + np.float32(0.5)
    target = rng.uniform(-3.0, 3.0, (n, 3)).astype(np.float32)
    target[:, 1] = np.abs(target[:, 1])
    d = (target - o).astype(np.float32)
    d[::97, 0] = 0.0                      # rays that take the select-form slab test
    d[5::211] = (0.0, 1.0, 0.0)           # straight up: mostly misses
    return o, d


def _free_bytes(hip):
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


@pytest.mark.parametrize("mode", [0, 1])
def test_noalloc_trace_equals_trace_rays_on_two_calls_in_a_row(oracle_lib, models_root, mode):
    from raytracing_amd import ComputeShader, _lib
    from raytracing_amd.core import RtError
    sc = oracle_lib.load_scene(models_root, 7)           # 21 levels deep: stacks reach the overflow part of the array
    cs = ComputeShader()
    setup_core(cs, sc, 16, 16, 1, 1)
    cs.setInt("traversalMode", mode)
    o1, d1 = _rays(_lib.NOALLOC_MAX_RAYS, 11)            # the maximum
    o2, d2 = _rays(1000, 12)                             # then a smaller input, not a multiple of 64

    with pytest.raises(RtError) as e:                    # the scene is not resident before the first allocating call
        cs.trace_rays_noalloc(o2, d2)
    assert e.value.status == 3
    want1 = cs.trace_rays(o1, d1)
    want2 = cs.trace_rays(o2, d2)
    assert (want1[0] >= 0).sum() > len(o1) // 4 and (want1[0] < 0).sum() > 0

    hip = ctypes.CDLL("libamdhip64.so")
    before = _free_bytes(hip)
    got1 = cs.trace_rays_noalloc(o1, d1)
    got2 = cs.trace_rays_noalloc(o2, d2)
    got3 = cs.trace_rays_noalloc(o1[:1], d1[:1])
    assert _free_bytes(hip) == before                    # nothing was allocated or freed on the device
    for got, want in ((got1, want1), (got2, want2), (got3, (want1[0][:1], want1[1][:1]))):
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(bits(got[1]), bits(want[1]))
    # and against the oracle's lineCast directly, on the small set
    tri, loc, _ = oracle_lib.linecast(sc, o2, d2)
    assert np.array_equal(got2[0], tri)
    hit = tri >= 0
    assert np.array_equal(bits(got2[1][hit]), bits(loc[hit]))

    with pytest.raises(RtError) as e:                    # beyond the array: refused, nothing traced
        cs.trace_rays_noalloc(np.concatenate([o1, o2]), np.concatenate([d1, d2]))
    assert e.value.status == 2 and "RT_NOALLOC_MAX_RAYS" in str(e.value)
    # a second context on the same GPU shares the array; used one after the other they do not disturb each other
    cs2 = ComputeShader()
    setup_core(cs2, oracle_lib.load_scene(models_root, 4), 16, 16, 1, 1)
    w4 = cs2.trace_rays(o2, d2)
    g4 = cs2.trace_rays_noalloc(o2, d2)
    assert np.array_equal(g4[0], w4[0]) and np.array_equal(bits(g4[1]), bits(w4[1]))
    again = cs.trace_rays_noalloc(o2, d2)
    assert np.array_equal(again[0], want2[0]) and np.array_equal(bits(again[1]), bits(want2[1]))
    # a new upload makes the scene non-resident again until an allocating call has run
    from raytracing_amd import SSBO
    SSBO.Bind(cs, sc.nodes, sc.nodes.nbytes, 3)
    with pytest.raises(RtError) as e:
        cs.trace_rays_noalloc(o2, d2)
    assert e.value.status == 3
    cs2.Delete()
    cs.Delete()
