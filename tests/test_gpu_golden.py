"""GPU: the HIP path against the committed known-answer vectors, with no oracle in the loop: the product's own
loader, builder and kernels through the C ABI must reproduce the golden images bit for bit (NaN == NaN), the
primary-hit triangle ids and the ray counts."""
import os

import numpy as np
import pytest

from tests.util import same_floats

pytestmark = pytest.mark.gpu
VEC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vectors")


# every execution strategy of the core must give the same bits: culled/ordered and reference-order traversal, the
# dynamic-fetch traversal kernels, one sample per pass on three overlapping streams, a single stream
VARIANTS = {"culled": {"traversalMode": 0}, "reference_order": {"traversalMode": 1}, "batch_kernels": {"refill": 0},
            "dynamic_fetch": {"refill": 1, "refillIdle": 24}, "dynamic_fetch_eager": {"refill": 1, "refillIdle": 1},
            "streams3_spp1": {"samplesPerPass": 1, "passStreams": 3}, "one_stream": {"samplesPerPass": 2, "passStreams": 1},
            # shadow rays of a bounce and closest-hit rays of the next in separate launches (the default traces them together),
            # and every pairing of batch / refill kernels in the joint launch
            "separate_launches": {"fusePairs": 0}, "pair_refill_batch": {"kernelChoice": 1}, "pair_batch_refill": {"kernelChoice": 2},
            "pair_refill_refill": {"kernelChoice": 3, "refillBlocksPerCU": 2, "blocksPerCU": 3},
            # two-level nodes in the refill kernels: joint launches with either or both kinds, separate launches, eager refills
            "wide_refill_refill": {"kernelChoice": 3, "wideNodes": 1}, "wide_refill_batch": {"kernelChoice": 1, "wideNodes": 1},
            "wide_batch_refill": {"kernelChoice": 2, "wideNodes": 1},
            "wide_separate": {"refill": 1, "refillIdle": 1, "fusePairs": 0, "wideNodes": 1},
            # bounce 0 in groups of 64 / 16 / 4 samples of 1 / 4 / 16 pixels per wave instead of one sample of 64 pixels
            "sample_groups_64": {"sampleGroup": 6}, "sample_groups_16_refill": {"sampleGroup": 4, "kernelChoice": 3},
            "sample_groups_4_spp3": {"sampleGroup": 2, "samplesPerPass": 3},
            # the contribution log in two parts: an entry per slot for the first bounces, per queue position for the later ones
            "late_log_after_2": {"logDenseBounces": 2, "lateLogPercent": 100}, "late_log_after_1_pairs": {"logDenseBounces": 1, "lateLogPercent": 100, "kernelChoice": 3},
            "late_log_after_3_batch": {"logDenseBounces": 3, "lateLogPercent": 100, "refill": 0, "fusePairs": 0, "samplesPerPass": 2, "passStreams": 2},
            # the sun's shadow rays traced a packet of 64 per wave (scalar loads, lane masks): every bounce, in joint and separate
            # launches, beside batch and refill kernels; and never (scenes 2 and 3 have a point light: the knob does not apply)
            "packet_all_bounces": {"packetBounces": 64}, "packet_separate": {"packetBounces": 64, "fusePairs": 0},
            "packet_beside_refill": {"packetBounces": 2, "kernelChoice": 3, "wideNodes": 1}, "packet_beside_batch": {"packetBounces": 1, "kernelChoice": 0},
            "packet_never": {"packetBounces": 0},
            # ... and a compact part too small for the frame: the core notices, renders the frame again with a dense log
            "late_log_overflow": {"logDenseBounces": 1, "lateLogPercent": 1}}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("scene", [2, 3, 4, 7])
def test_hip_path_reproduces_golden(models_root, scene, variant):
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    g = np.load(os.path.join(VEC, "scene%d.npz" % scene), allow_pickle=False)
    _, W, H, spp, b = [int(v) for v in g["config"]]
    cs = ComputeShader()
    S.Scene.LoadScene(cs, scene, models_root)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", b)
    for k, v in VARIANTS[variant].items():
        cs.setInt(k, v)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    cs.dispatch()
    for k, name in enumerate(("color", "albedo", "normal", "depth")):
        assert same_floats(cs.get_image(k), g[name]).all(), name
    tri, hit = cs.primary_hits()
    assert np.array_equal(tri, g["prim_tri"])
    assert same_floats(hit, g["prim_hit"]).all()
    st = cs.stats()
    rp, rb, rs = [int(v) for v in g["rays"]]
    assert st["rays_primary"] == rp and st["rays_bounce"] == rb
    # shadow casts the reference issues = traced + skipped (N.L <= 0: the result cannot matter)
    assert st["rays_shadow"] + st["rays_shadow_skipped"] == rs
    if variant.startswith("late_log"):
        overflow = variant == "late_log_overflow"
        assert st["log_fallbacks"] == (1 if overflow else 0), st
        assert st["log_dense_bounces"] == (b if overflow else VARIANTS[variant]["logDenseBounces"]), st
    else:
        assert st["log_dense_bounces"] == b and st["log_fallbacks"] == 0, st      # 8 bounces or fewer: the whole log is dense
    cs.Delete()
