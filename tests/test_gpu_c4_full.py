"""GPU: BASELINE configs[3] at its real size — the synthetic height field of 2 x 3162^2 = 19 996 488 triangles that
bench.py --workload c4 times — checked, once.  It is the only configuration whose buffers cross 2 GiB (2.56 GB of
Triangle records through one 32-bit-sized SSBO::Bind, 48-byte leaf codes first*8+count up to 1.6e8) and whose tree is
deeper than the kernels' LDS stack part by ~20 levels, so the global stack overflow area carries real traffic.

(i)  1920x1080, 8 bounces: the default kernels (for a BVH this size: the refill kernels on two-level nodes), the batch
     kernels (refill = 0), the refill kernels on the binary nodes (wideNodes = 0) and the reference-order exhaustive
     walk (traversalMode = 1) give bit-identical images (all four), primary-hit ids and ray counts.
(ii) The oracle (oracle/rt_ref.cpp: lineCast and the full integrator, shader.comp:243-537) runs on the very buffers the
     product's loader and builder bound (rth_scene_buffer): a few thousand rt_trace_rays rays and two 12x8 windows of
     the full-size frame, bit for bit.  The oracle does not rebuild the BVH here: builder parity is pinned at small sizes
     (tests/test_oracle_bvh.py, test_host_loader.py); what is checked here is traversal + shading at this size."""
import numpy as np
import pytest

from tests.util import bits, same_floats

pytestmark = pytest.mark.gpu

GRID = 3162
W, H, SPP, BOUNCES = 1920, 1080, 2, 8
WINDOWS = [(954, 300, 12, 8), (400, 515, 12, 8)]      # (x, y, w, h): the near field under the camera; rays skimming the field just below the horizon


def test_c4_full_size_all_kernel_families_and_oracle_windows(oracle_lib, tmp_path):
    from oracle import gltf_flatten as G
    from raytracing_amd import ComputeShader, synthetic
    from raytracing_amd import scene as S
    path = str(tmp_path / "hf20m")
    ntri, nvert = synthetic.write_gltf(path, GRID)
    assert ntri == 2 * GRID * GRID and nvert <= 1 << 24
    cs = ComputeShader()
    S.Scene.LoadModels(cs, [path + ".gltf"], synthetic.SYNTHETIC_LIGHTS)
    print("\nc4 full size: %d triangles, %d vertices, load %s" % (ntri, nvert, S.Scene.timings()))
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", SPP)
    cs.setInt("maxBounces", BOUNCES)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)

    # (i) three kernel families, one answer
    runs = {}
    for name, knobs in (("default", {}), ("batch", {"refill": 0}), ("binary nodes", {"wideNodes": 0}), ("reference order", {"traversalMode": 1})):
        for k, v in knobs.items():
            cs.setInt(k, v)
        cs.dispatch()
        st = cs.stats()
        print("  %-16s %.0f ms, rays %d / %d / %d (+%d skipped)" % (name, st["ms_total"], st["rays_primary"], st["rays_bounce"], st["rays_shadow"], st["rays_shadow_skipped"]))
        runs[name] = ([cs.get_image(k) for k in range(4)], cs.primary_hits()[0], st)
        for k in knobs:
            cs.setInt(k, {"refill": -1, "traversalMode": 0, "wideNodes": -1}[k])
    base = runs["default"]
    assert base[2]["wide_nodes"] == 1 and runs["binary nodes"][2]["wide_nodes"] == 0 and base[2]["wide_depth"] >= 10
    assert base[2]["rays_primary"] == W * H * SPP
    assert (base[1] >= 0).mean() > 0.3                     # the field fills the lower part of the frame
    assert int(base[1].max()) > 1 << 24                    # ids beyond what a float index could name
    for name in ("batch", "binary nodes", "reference order"):
        other = runs[name]
        for k in range(4):
            assert same_floats(other[0][k], base[0][k]).all(), (name, k)
        assert np.array_equal(other[1], base[1]), name
        for key in ("rays_primary", "rays_bounce", "rays_shadow", "rays_shadow_skipped"):
            assert other[2][key] == base[2][key], (name, key)

    # (ii) the oracle on the product-built buffers
    flat = {
        "vertices": np.frombuffer(S.Scene.buffer_view(0), G.VERTEX_DT),
        "matrices": np.frombuffer(S.Scene.buffer_view(1), np.float32).reshape(-1, 4, 4),
        "materials": np.frombuffer(S.Scene.buffer_view(2), G.MATERIAL_DT),
        "triangles": np.frombuffer(S.Scene.buffer_view(4), G.TRIANGLE_DT),
        "lights": np.frombuffer(S.Scene.buffer_view(5), G.LIGHT_DT).copy(),
        "hasSun": 1, "textures": [],
    }
    nodes = np.frombuffer(S.Scene.buffer_view(3), G.NODE_DT)
    assert len(flat["triangles"]) == ntri and flat["triangles"].nbytes > 1 << 31
    assert len(nodes) > ntri // 2
    sc = oracle_lib.Scene(flat, nodes=nodes, textures=[])

    rng = np.random.default_rng(2024)
    n = 4096
    o = np.empty((n, 3), np.float32)
    o[:, 0] = rng.uniform(-13, 13, n); o[:, 2] = rng.uniform(-13, 13, n); o[:, 1] = rng.uniform(0.2, 9.0, n)
    target = np.stack([rng.uniform(-12, 12, n), rng.uniform(-1.0, 3.0, n), rng.uniform(-12, 12, n)], -1)
    d = (target - o).astype(np.float32)
    d[::53, 1] = 0.0                                       # skimming rays with a zero component (select-form slab test)
    d[7::101] = (0.0, -1.0, 0.0)                           # straight down: two zero components
    want_tri, want_hit, _ = oracle_lib.linecast(sc, o, d)
    assert (want_tri >= 0).sum() > n // 2
    print("  oracle lineCast: %d of %d rays hit, %d nodes" % ((want_tri >= 0).sum(), n, len(nodes)))
    for mode in (0, 1):
        cs.setInt("traversalMode", mode)
        tri, hit = cs.trace_rays(o, d)
        assert np.array_equal(tri, want_tri), mode
        m = want_tri >= 0
        assert np.array_equal(bits(hit[m]), bits(want_hit[m])), mode
    cs.setInt("traversalMode", 0)

    for (x, y, w, h) in WINDOWS:
        p = oracle_lib.make_params(sc, W, H, SPP, BOUNCES, math_mode=1, region=(x, y, x + w, y + h))
        want = oracle_lib.render(sc, p)
        for k, name in enumerate(("color", "albedo", "normal", "depth")):
            assert same_floats(base[0][k][y:y + h, x:x + w], want[name][y:y + h, x:x + w]).all(), (x, y, name)
        assert np.array_equal(base[1][y:y + h, x:x + w], want["prim_tri"][y:y + h, x:x + w])
    assert (base[1][300:308, 954:966] >= 0).all()          # the first window really looks at the field
    cs.Delete()
