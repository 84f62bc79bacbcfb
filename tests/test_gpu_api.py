"""GPU: behaviour of the C ABI around the hot path — regions / sample ranges (the multi-GPU partition), invariance to
the number of samples in flight, ragged image sizes, error reporting, the host loader on the GPU, 8-bit frames."""
import numpy as np
import pytest

from tests.util import bits, same_floats, setup_core

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene4(oracle_lib, models_root):
    return oracle_lib.load_scene(models_root, 4)


@pytest.fixture(scope="module")
def scene7(oracle_lib, models_root):
    return oracle_lib.load_scene(models_root, 7)


def _render(cs, region=None, spp_range=None):
    if region is None and spp_range is None:
        cs.dispatch()
    else:
        x, y, w, h = region or (0, 0, cs.width, cs.height)
        r0, r1 = spp_range
        cs.dispatch_region(x, y, w, h, r0, r1)
    return [cs.get_image(k) for k in range(4)]


def test_ragged_size_and_single_sample(oracle_lib, scene4):
    """Sizes that are not multiples of the 8x8 ray tile (dead slots), spp = 1, 1 bounce."""
    from raytracing_amd import ComputeShader
    for W, H, spp, b in ((50, 37, 3, 4), (9, 5, 1, 1), (64, 8, 2, 2)):
        want = oracle_lib.render(scene4, oracle_lib.make_params(scene4, W, H, spp, b, math_mode=1))
        cs = ComputeShader()
        setup_core(cs, scene4, W, H, spp, b)
        imgs = _render(cs)
        for k, name in enumerate(("color", "albedo", "normal", "depth")):
            assert same_floats(imgs[k], want[name]).all(), (W, H, name)
        st = cs.stats()
        assert st["rays_primary"] == W * H * spp
        cs.Delete()


def test_samples_per_pass_does_not_change_the_image(oracle_lib, scene7):
    from raytracing_amd import ComputeShader
    W, H, spp, b = 64, 40, 6, 8
    cs = ComputeShader()
    setup_core(cs, scene7, W, H, spp, b)
    ref_img = None
    for S in (1, 2, 4, 6, 32):
        cs.setInt("samplesPerPass", S)
        img = _render(cs)[0]
        if ref_img is None:
            ref_img = img
            want = oracle_lib.render(scene7, oracle_lib.make_params(scene7, W, H, spp, b, math_mode=1))
            assert same_floats(img, want["color"]).all()
        else:
            assert same_floats(img, ref_img).all(), S
    cs.Delete()


def test_sample_groups_and_pass_shapes_do_not_change_the_image(oracle_lib, scene7):
    """Slot numbering (sampleGroup: 1 .. 64 samples of a pixel per wave), ragged and evened-out passes (13 samples over 3
    streams, 4 + 4 + 4 + 1 or as given) and a path-state budget so small that it sets the pass size: all four images, the
    primary ids and the ray counts are those of the oracle-checked default."""
    from raytracing_amd import ComputeShader
    W, H, spp, b = 72, 40, 13, 8          # 72 x 40: nine 8 x 8 tile columns, the last pixel group of a row of tiles is ragged for no setting
    cs = ComputeShader()
    setup_core(cs, scene7, W, H, spp, b)
    first = _render(cs)
    want = oracle_lib.render(scene7, oracle_lib.make_params(scene7, W, H, spp, b, math_mode=1))
    for k, name in enumerate(("color", "albedo", "normal", "depth")):
        assert same_floats(first[k], want[name]).all(), name
    tri0, hit0 = cs.primary_hits()
    rays0 = [cs.stats()[k] for k in ("rays_primary", "rays_bounce", "rays_shadow", "rays_shadow_skipped")]
    defaults = {"sampleGroup": -1, "samplesPerPass": 64, "passStreams": 0, "autoPass": 1, "pathMemPercent": 75, "kernelChoice": -1,
                "logDenseBounces": 0, "lateLogPercent": 25}
    for knobs in ({"sampleGroup": 0}, {"sampleGroup": 6}, {"sampleGroup": 3, "samplesPerPass": 12}, {"sampleGroup": 2, "samplesPerPass": 4, "passStreams": 3},
                  {"sampleGroup": 6, "samplesPerPass": 5, "autoPass": 0}, {"sampleGroup": 1, "samplesPerPass": 3, "passStreams": 2, "kernelChoice": 3},
                  {"pathMemPercent": 1, "sampleGroup": 2}, {"logDenseBounces": 3, "lateLogPercent": 100, "sampleGroup": 6},
                  {"logDenseBounces": 2, "lateLogPercent": 100, "samplesPerPass": 4, "passStreams": 3},
                  {"sampleGroup": 6, "logDenseBounces": 2, "lateLogPercent": 100}):
        for k, v in knobs.items():
            cs.setInt(k, v)
        imgs = _render(cs)
        for k in range(4):
            assert same_floats(imgs[k], first[k]).all(), (knobs, k)
        tri, hit = cs.primary_hits()
        assert np.array_equal(tri, tri0) and same_floats(hit, hit0).all(), knobs
        assert [cs.stats()[k] for k in ("rays_primary", "rays_bounce", "rays_shadow", "rays_shadow_skipped")] == rays0, knobs
        for k in knobs:
            cs.setInt(k, defaults[k])
    cs.Delete()


def test_camera_moves_between_frames(oracle_lib, scene7):
    """Three camera positions in a row on one context, each frame equal to the oracle's, then back to the first: nothing
    derived from the camera is kept across frames."""
    from raytracing_amd import ComputeShader
    from tests.util import set_camera
    from oracle import ref
    W, H, spp, b = 64, 40, 4, 6
    cs = ComputeShader()
    setup_core(cs, scene7, W, H, spp, b)
    first = None
    cams = [ref.DEFAULT_CAMERA, ((2.5, 1.5, 4.0), ref.DEFAULT_CAMERA[1]), ((-3.0, 2.0, 3.5), ref.DEFAULT_CAMERA[1]), ref.DEFAULT_CAMERA]
    for k, cam in enumerate(cams):
        set_camera(cs, cam, W, H)
        imgs = _render(cs)
        want = oracle_lib.render(scene7, oracle_lib.make_params(scene7, W, H, spp, b, math_mode=1, camera=cam))
        for i, name in enumerate(("color", "albedo", "normal", "depth")):
            assert same_floats(imgs[i], want[name]).all(), (k, name)
        if k == 0:
            first = imgs
        if k == 3:
            assert all(same_floats(imgs[i], first[i]).all() for i in range(4))
    cs.Delete()


def test_a_context_that_grows_and_shrinks(oracle_lib, scene4):
    """One context, no new upload: a small frame, a frame 12 times the pixels with ten times the samples and twice the
    bounces (the path buffers — one arena per array, sized for both pass shapes — are allocated again), the same with the
    other pass shape pinned, and the small frame once more.  Every frame is the oracle's; the last equals the first."""
    from raytracing_amd import ComputeShader
    from tests.util import set_camera
    from oracle import ref
    cs = ComputeShader()
    setup_core(cs, scene4, 48, 40, 4, 4)
    shots = {}
    for step, (W, H, spp, b, choice) in enumerate(((48, 40, 4, 4, -1), (200, 120, 40, 8, -1), (200, 120, 40, 8, 5), (48, 40, 4, 4, -1))):
        cs.resize(W, H)
        set_camera(cs, ref.DEFAULT_CAMERA, W, H)
        cs.setInt("samplesPerPixel", spp)
        cs.setInt("maxBounces", b)
        cs.setInt("kernelChoice", choice)
        imgs = _render(cs)
        if (W, H) not in shots:
            want = oracle_lib.render(scene4, oracle_lib.make_params(scene4, W, H, spp, b, math_mode=1))
            for k, name in enumerate(("color", "albedo", "normal", "depth")):
                assert same_floats(imgs[k], want[name]).all(), (step, name)
            shots[(W, H)] = imgs
        else:
            assert all(same_floats(imgs[k], shots[(W, H)][k]).all() for k in range(4)), step
        assert cs.stats()["rays_primary"] == W * H * spp
    cs.Delete()


def test_region_is_bit_identical_and_sample_ranges_sum(oracle_lib, scene4):
    from raytracing_amd import ComputeShader
    W, H, spp, b = 64, 48, 4, 6
    cs = ComputeShader()
    setup_core(cs, scene4, W, H, spp, b)
    full = _render(cs)
    # row band [16, 40): identical inside, zero outside, all four images
    band = _render(cs, region=(0, 16, W, 24), spp_range=(0, spp))
    for k in range(4):
        assert same_floats(band[k][16:40], full[k][16:40]).all()
        assert (band[k][:16] == 0).all() and (band[k][40:] == 0).all()
    # a window not aligned to the tile grid
    win = _render(cs, region=(5, 3, 30, 21), spp_range=(0, spp))
    assert same_floats(win[0][3:24, 5:35], full[0][3:24, 5:35]).all()
    # sample ranges: colour sums to the full frame up to float re-association; AOVs only from the range holding r = 0
    a = _render(cs, spp_range=(0, 1))
    c = _render(cs, spp_range=(1, spp))
    assert np.allclose(a[0][..., :3] + c[0][..., :3], full[0][..., :3], rtol=1e-5, atol=1e-6)
    for k in (1, 2, 3):
        assert same_floats(a[k], full[k]).all() and (c[k] == 0).all()
    # alpha: 1 from the range holding r = 0, 0 from the others, so the merged frame has the shader's vec4(rgb, 1)
    assert (a[0][..., 3] == 1).all() and (c[0][..., 3] == 0).all() and (full[0][..., 3] == 1).all()
    # empty sample range: clears, traces nothing
    e = _render(cs, spp_range=(2, 2))
    assert (e[0][..., :3] == 0).all() and cs.stats()["rays_primary"] == 0
    cs.Delete()


def test_zero_bounces_and_no_lights(oracle_lib, scene4):
    from raytracing_amd import ComputeShader
    W, H = 32, 24
    cs = ComputeShader()
    setup_core(cs, scene4, W, H, 2, 0)
    img = _render(cs)
    assert (img[0][..., :3] == 0).all() and (img[0][..., 3] == 1).all()      # maxBounces = 0: the loop body never runs
    assert (img[2][..., :3] == 0.5).all()
    cs.setInt("maxBounces", 3)
    cs.setInt("numLights", 0)
    got = _render(cs)[0]
    p = oracle_lib.make_params(scene4, W, H, 2, 3, math_mode=1)
    p.numLights = 0
    want = oracle_lib.render(scene4, p)
    assert same_floats(got, want["color"]).all()
    assert cs.stats()["rays_shadow"] == 0
    cs.Delete()


def test_errors_are_reported_not_crashes(scene4):
    from raytracing_amd import ComputeShader, SSBO
    from raytracing_amd.core import RtError
    cs = ComputeShader()
    with pytest.raises(RtError) as e:
        cs.dispatch()
    assert e.value.status == 3                      # RT_ERR_NOT_READY: no image size
    cs.resize(16, 16)
    with pytest.raises(RtError) as e:
        cs.dispatch()
    assert e.value.status == 3                      # slots missing
    with pytest.raises(RtError) as e:
        SSBO.Bind(cs, np.zeros(10, np.uint8), 10, 3)
    assert e.value.status == 2                      # not a multiple of 48
    with pytest.raises(RtError):
        SSBO.Bind(cs, np.zeros(48, np.uint8), 48, 9)
    setup_core(cs, scene4, 16, 16, 1, 1)
    bad = scene4.triangles.copy()
    bad["indices"][0, 0] = 10 ** 7
    SSBO.Bind(cs, bad, bad.nbytes, 4)
    with pytest.raises(RtError) as e:
        cs.dispatch()
    assert e.value.status == 5                      # RT_ERR_SCENE: vertex index out of range, caught on the host
    nodes = scene4.nodes.copy()
    nodes["children"][0, 1] = 0                     # cycle
    SSBO.Bind(cs, scene4.triangles, scene4.triangles.nbytes, 4)
    SSBO.Bind(cs, nodes, nodes.nbytes, 3)
    with pytest.raises(RtError) as e:
        cs.dispatch()
    assert e.value.status == 5
    with pytest.raises(RtError):
        cs.dispatch_region(0, 0, 17, 16, 0, 1)      # region outside the image
    for bad_region in ((0xFFFFFFFF, 0, 2, 1), (0, 0xFFFFFFFF, 1, 2), (15, 0, 0xFFFFFFF2, 1), (16, 0, 1, 1), (0, 0, 0, 1)):
        with pytest.raises(RtError) as e:           # x + w wraps in 32 bits: still refused
            cs.dispatch_region(*bad_region, 0, 1)
        assert e.value.status == 2
    cs.setInt("projection", 1)                      # unknown uniform names are ignored like GL location -1
    cs.setMat4("invView", np.eye(4))
    cs.Delete()


@pytest.mark.parametrize("scene_id", [2, 4, 7])
def test_host_loader_end_to_end(oracle_lib, models_root, scene_id):
    """Scene::LoadScene through the C++ host layer on the GPU (textures included) equals the oracle-side set-up."""
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    W, H, spp, b = 80, 48, 2, 6
    cs = ComputeShader()
    S.Scene.LoadScene(cs, scene_id, models_root)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", b)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    cs.dispatch()
    sc = oracle_lib.load_scene(models_root, scene_id)
    want = oracle_lib.render(sc, oracle_lib.make_params(sc, W, H, spp, b, math_mode=1))
    assert same_floats(cs.get_image(0), want["color"]).all()
    # Renderer::NextFrame's 8-bit conversion: clamp, truncate, vertical flip
    rgba8 = S.frame_rgba8(cs)
    f = np.nan_to_num(np.clip(want["color"][::-1], 0, 1), nan=0.0)
    assert np.array_equal(rgba8, (f * np.float32(255)).astype(np.uint8))
    # orbit camera (config 5's path) through the host helper renders and differs from the default view
    pos, ori = S.orbit_camera(30)
    S.set_camera(cs, pos, ori, W, H, 45.0)
    cs.dispatch()
    cam = (pos, ori)
    want2 = oracle_lib.render(sc, oracle_lib.make_params(sc, W, H, spp, b, camera=cam, math_mode=1))
    assert same_floats(cs.get_image(0), want2["color"]).all()
    cs.Delete()


def test_write_image_and_device_pointer(scene4):
    from raytracing_amd import ComputeShader
    cs = ComputeShader()
    setup_core(cs, scene4, 16, 8, 1, 1)
    img = np.random.default_rng(0).random((8, 16, 4), dtype=np.float32)
    cs.set_image(0, img)                             # glTexSubImage2D write-back (main.cpp:184)
    assert np.array_equal(cs.get_image(0), img)
    assert cs.image_device_ptr(0) != 0 and cs.image_device_ptr(3) != cs.image_device_ptr(0)
    cs.Delete()


def test_denoiser_hand_off_rgb_round_trip(scene4):
    """rt_read_image_rgb / rt_write_image_rgb = the reference's rgbaTOrgb read-back into OIDN's Float3 buffers and the
    alpha-1 write-back of the denoised colour (main.cpp:152-184), repacked on the GPU."""
    from raytracing_amd import ComputeShader
    W, H = 70, 41                                    # not a multiple of anything
    cs = ComputeShader()
    setup_core(cs, scene4, W, H, 2, 3)
    cs.dispatch()
    for k in range(4):
        assert np.array_equal(bits(cs.get_image_rgb(k)), bits(cs.get_image(k)[..., :3]))
    rng = np.random.default_rng(5)
    den = rng.random((H, W, 3), dtype=np.float32)
    cs.set_image_rgb(0, den)
    back = cs.get_image(0)
    assert np.array_equal(back[..., :3], den) and (back[..., 3] == 1).all()
    assert np.array_equal(cs.get_image_rgb(0), den)
    cs.Delete()


def test_frame_rgba8_is_the_reference_conversion(scene4):
    """rt_read_frame_rgba8 (through rth_frame_rgba8) = Renderer::NextFrame's clamp / * 255 / truncate and its vertical
    flip (Renderer.cpp:95-127), converted on the GPU; a size that is not the image's is refused, not overrun."""
    import ctypes as C
    from raytracing_amd import ComputeShader, _lib
    from raytracing_amd import scene as S
    W, H = 90, 52
    cs = ComputeShader()
    setup_core(cs, scene4, W, H, 3, 4)
    cs.dispatch()
    img = cs.get_image(0)
    img[5, 7] = (np.nan, -0.5, 7.0, 0.999)                     # NaN -> 0, clamps, truncation
    cs.set_image(0, img)
    got = S.frame_rgba8(cs)
    f = np.where(np.isnan(img), np.float32(0), np.clip(img, np.float32(0), np.float32(1)))
    want = (f * np.float32(255)).astype(np.uint8)[::-1]
    assert np.array_equal(got, want)
    assert tuple(got[H - 1 - 5, 7]) == (0, 0, 255, 254)
    small = np.empty(16, np.uint8)
    st = _lib.load().rt_read_frame_rgba8(cs._h, small.ctypes.data_as(C.c_void_p), C.c_uint64(small.nbytes))
    assert st == 2                                            # RT_ERR_INVALID
    cs.Delete()


def test_camera_path_driver_writes_reference_frames(oracle_lib, models_root, tmp_path):
    """Config 5's driver (raytracing_amd/render_path.py = Renderer::Start/NextFrame/UpdateCamera): frame 0 with the
    default camera, frame f >= 1 with the orbit camera of UpdateCamera(currentFrame = f - 1), 8-bit PPM frames as Renderer.cpp:103-124 converts them."""
    from raytracing_amd import render_path
    W, H, spp, b = 64, 36, 2, 4
    render_path.main(["--scene", "4", "--models", models_root, "--width", str(W), "--height", str(H), "--spp", str(spp),
                      "--bounces", str(b), "--frames", "3", "--out", str(tmp_path)])
    d = tmp_path / ("Scene-4_spp-%d_bounces-%d" % (spp, b))
    sc = oracle_lib.load_scene(models_root, 4)
    for f in range(3):
        raw = (d / ("frame_%04d.ppm" % f)).read_bytes()
        hdr = b"P6\n%d %d\n255\n" % (W, H)
        assert raw.startswith(hdr)
        got = np.frombuffer(raw[len(hdr):], np.uint8).reshape(H, W, 3)
        cam = oracle_lib.DEFAULT_CAMERA if f == 0 else render_path.frame_camera(f)
        want = oracle_lib.render(sc, oracle_lib.make_params(sc, W, H, spp, b, camera=cam, math_mode=1))["color"]
        ref8 = (np.nan_to_num(np.clip(want[::-1, :, :3], 0, 1), nan=0.0) * np.float32(255)).astype(np.uint8)
        assert np.array_equal(got, ref8), f


def test_cpp_example_renders_the_reference_frames(oracle_lib, models_root, tmp_path):
    """examples/headless_main.cpp (CMake build, C++ host classes with the reference's names over the C ABI): the offline
    render writes frame 0 with the default camera and the following frames on the orbit, as Renderer.cpp does."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "headless_main")
    if not os.path.exists(exe):
        pytest.skip("build/headless_main is not built (cmake -S . -B build -G Ninja && cmake --build build)")
    W, H, spp, b = 64, 36, 2, 4
    out = subprocess.run([exe, models_root, "4", str(W), str(H), str(spp), str(b), "2", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    d = tmp_path / ("Scene-4_spp-%d_bounces-%d" % (spp, b))
    sc = oracle_lib.load_scene(models_root, 4)
    raw = (d / "frame_0000.ppm").read_bytes()
    hdr = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(hdr)
    got = np.frombuffer(raw[len(hdr):], np.uint8).reshape(H, W, 3)
    want = oracle_lib.render(sc, oracle_lib.make_params(sc, W, H, spp, b, math_mode=1))["color"]
    ref8 = (np.nan_to_num(np.clip(want[::-1, :, :3], 0, 1), nan=0.0) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(got, ref8)
    assert (d / "frame_0001.ppm").exists()


def test_cpp_example_on_two_contexts_writes_the_same_frames(models_root, tmp_path):
    """examples/headless_main.cpp with its ninth argument ("0,0": two contexts on the box's one GPU): the C++ multi-GPU frame
    driver (rt_host::MultiGPU; sample ranges, one merge per frame) writes the frames of the offline render; 8-bit frames of
    a sample-split render may differ from the single-context ones by a unit in a few pixels (float re-association)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "headless_main")
    if not os.path.exists(exe):
        pytest.skip("build/headless_main is not built (cmake -S . -B build -G Ninja && cmake --build build)")
    W, H, spp, b = 64, 36, 4, 4
    frames = {}
    for tag, extra in (("one", []), ("two", ["0,0"])):
        d0 = tmp_path / tag
        d0.mkdir()
        out = subprocess.run([exe, models_root, "4", str(W), str(H), str(spp), str(b), "2", str(d0)] + extra, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stdout + out.stderr
        if extra:
            assert "on 2 contexts" in out.stdout
        d = d0 / ("Scene-4_spp-%d_bounces-%d" % (spp, b))
        hdr = b"P6\n%d %d\n255\n" % (W, H)
        frames[tag] = [np.frombuffer((d / ("frame_%04d.ppm" % f)).read_bytes()[len(hdr):], np.uint8).astype(np.int32) for f in range(2)]
    for f in range(2):
        assert np.abs(frames["one"][f] - frames["two"][f]).max() <= 1


def test_full_size_frame_is_reproducible_and_stream_independent(oracle_lib, scene4):
    """1920x1080, 48 spp: two renders give the same bits, and so do one stream, three streams with ragged passes and the
    default — the per-pixel sums are replayed in sample order whatever overlaps (the size the multi-stream passes are for)."""
    from raytracing_amd import ComputeShader
    W, H, spp, b = 1920, 1080, 48, 8
    cs = ComputeShader()
    setup_core(cs, scene4, W, H, spp, b)
    cs.dispatch()
    first = cs.get_image(0).copy()
    st = cs.stats()
    assert st["rays_primary"] == W * H * spp
    cs.dispatch()
    assert np.array_equal(bits(cs.get_image(0)), bits(first))
    for knobs in ({"passStreams": 1}, {"passStreams": 3, "samplesPerPass": 7, "autoPass": 0}, {"refill": 1}, {"blocksPerCU": 5},
                  {"kernelChoice": 2}, {"kernelChoice": 1}, {"kernelChoice": 5}):      # (5: bit 2 = half the streams, twice the samples per pass)
        for k, v in knobs.items():
            cs.setInt(k, v)
        cs.dispatch()
        assert np.array_equal(bits(cs.get_image(0)), bits(first)), knobs
        if "kernelChoice" in knobs:      # the pinned combination is the one that ran, and is reported back (multi-GPU jobs pin rank 0's)
            assert cs.stats()["kernel_choice"] == knobs["kernelChoice"] and cs.stats()["kernel_choice_settled"] == 1
        for k in knobs:
            cs.setInt(k, {"passStreams": 0, "samplesPerPass": 64, "autoPass": 1, "refill": -1, "blocksPerCU": 64, "kernelChoice": -1}[k])
    cs.Delete()


def test_small_batch_grid_with_refill_kernels_keeps_stacks_apart(models_root):
    """blocksPerCU = 1 shrinks the batch kernels' grid below the refill kernels' (4 to 6 workgroups per CU): the stack
    overflow area is sized and strided for the larger of the two, so deep stacks (scene 7: 21 levels, 8 in LDS) of
    different threads never alias.  Frame of 0.9 M slots per pass, far more than either grid."""
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    W, H, spp = 640, 360, 4
    cs = ComputeShader()
    S.Scene.LoadScene(cs, 7, models_root)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", 8)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    cs.setInt("refill", 0)
    cs.dispatch()
    first = cs.get_image(0).copy()
    for knobs in ({"refill": 1, "blocksPerCU": 1}, {"refill": 1, "blocksPerCU": 2, "refillBlocksPerCU": 9}, {"refill": 1, "passStreams": 1, "blocksPerCU": 1}):
        for k, v in knobs.items():
            cs.setInt(k, v)
        cs.dispatch()
        assert np.array_equal(bits(cs.get_image(0)), bits(first)), knobs
        for k in knobs:
            cs.setInt(k, {"passStreams": 0, "refill": 0, "blocksPerCU": 64, "refillBlocksPerCU": 0}[k])
    cs.Delete()


@pytest.mark.parametrize("W,H,spp", [(7680, 4320, 16), (15360, 8640, 4), (1, 1, 64), (3, 4096, 5), (16384, 2, 3)])
def test_extreme_frame_shapes_agree_across_kernel_families(models_root, W, H, spp):
    """Scene 7 (glass, 16 k triangles) from a single pixel to a 16K frame of 133 M pixels, and one-pixel-wide strips: the
    default path, the refill kernels and one stream with one sample per pass give the same bits, every primary ray is
    cast, and the only non-finite pixels are NaN (the reference's own 0/0) in all four channels but
    alpha."""
    from raytracing_amd import ComputeShader
    from raytracing_amd import scene as S
    cs = ComputeShader()
    S.Scene.LoadScene(cs, 7, models_root)
    cs.resize(W, H)
    cs.setInt("samplesPerPixel", spp)
    cs.setInt("maxBounces", 8)
    S.set_camera(cs, S.DEFAULT_CAMERA[0], S.DEFAULT_CAMERA[1], W, H, 45.0)
    first = None
    for knobs in ({}, {"refill": 1}, {"passStreams": 1, "samplesPerPass": 1}):
        for k, v in knobs.items():
            cs.setInt(k, v)
        cs.dispatch_region(0, 0, W, H, 0, spp)
        img = cs.get_image(0)
        assert cs.stats()["rays_primary"] == W * H * spp
        if first is None:
            first = img.copy()
            assert not np.isinf(first).any()
            assert (first[..., 3] == 1.0).all()
        else:
            assert np.array_equal(bits(img), bits(first)), knobs
        for k in knobs:
            cs.setInt(k, {"passStreams": 0, "samplesPerPass": 64, "refill": -1}[k])
    cs.Delete()


def test_materials_that_gain_textures_after_the_first_frame(oracle_lib, models_root):
    """The per-triangle texture-coordinate table is built and uploaded only once a material samples a texture: scene 3
    first with its texture flags cleared (no table), then with the real materials bound to slot 2 alone — the second frame
    must be the textured scene's, bit for bit, and a third upload of the plain materials the plain frame again."""
    from raytracing_amd import ComputeShader, SSBO
    sc = oracle_lib.load_scene(models_root, 3)
    assert (sc.materials["hasBaseTexture"] == 1).any()
    W, H, spp, b = 96, 64, 2, 4
    want_tex = oracle_lib.render(sc, oracle_lib.make_params(sc, W, H, spp, b, math_mode=1))
    import copy
    plain = copy.copy(sc)
    plain.materials = sc.materials.copy()
    for fld in ("hasBaseTexture", "hasNormalTexture", "hasMetallicRoughnessTexture"):
        plain.materials[fld] = 0
    want_plain = oracle_lib.render(plain, oracle_lib.make_params(plain, W, H, spp, b, math_mode=1))
    assert not same_floats(want_tex["color"], want_plain["color"]).all()
    cs = ComputeShader()
    setup_core(cs, sc, W, H, spp, b)                      # creates the textures, binds the textured materials ...
    SSBO.Bind(cs, plain.materials, plain.materials.nbytes, 2)      # ... which the plain ones replace before the first frame
    cs.dispatch()
    assert same_floats(cs.get_image(0), want_plain["color"]).all()
    mats = sc.materials.copy()                              # handles 1.. are the textures in creation order (setup_core)
    SSBO.Bind(cs, mats, mats.nbytes, 2)
    cs.dispatch()
    for k, name in enumerate(("color", "albedo", "normal", "depth")):
        assert same_floats(cs.get_image(k), want_tex[name]).all(), name
    SSBO.Bind(cs, plain.materials, plain.materials.nbytes, 2)
    cs.dispatch()
    assert same_floats(cs.get_image(0), want_plain["color"]).all()
    cs.Delete()


def test_contexts_release_their_device_memory(scene4):
    """rt_destroy (ComputeShader::Delete) frees images, scene, path buffers, streams and events: free device memory after
    twelve create / render / destroy rounds is what it was after the first."""
    import ctypes
    from raytracing_amd import ComputeShader
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    def one_round():
        cs = ComputeShader()
        setup_core(cs, scene4, 256, 144, 8, 4)
        cs.dispatch()
        cs.setInt("timeKernels", 1)
        cs.dispatch_region(0, 0, 256, 144, 0, 8)
        cs.Delete()

    one_round()
    before = free_bytes()
    for _ in range(12):
        one_round()
    assert free_bytes() >= before - (64 << 20)      # allocator granularity, not a per-round loss


def test_settled_kernel_choice_is_what_the_next_frame_runs(scene4):
    """on the frame that finishes the per-scene trial, rt_stats must name the combination the trial kept (what the
    NEXT frame runs), not the last trial's own combination — distributed.agree_on_kernels pins exactly that number on every
    rank.  Frames of 1280 x 720 x 16 samples (>= 1e7 samples, >= 3e7 rays) qualify as trial frames."""
    from raytracing_amd import ComputeShader
    cs = ComputeShader()
    setup_core(cs, scene4, 1280, 720, 16, 8)
    cs.setInt("tuneCache", 0)                              # (an earlier test of this process may have settled this scene already)
    seen = []
    for _ in range(10):
        cs.dispatch()
        st = cs.stats()
        seen.append((st["kernel_choice"], st["kernel_choice_settled"]))
    settled = [i for i, (_, s) in enumerate(seen) if s]
    assert settled, seen                                   # the trial finishes within a few frames
    first = settled[0]
    assert first >= 4, seen                                # four trial frames (frame 0 allocates and does not count)
    assert all(s for _, s in seen[first:]), seen
    assert len({c for c, _ in seen[first:]}) == 1, seen    # the settling frame already reports the kept choice
    # pinned by a knob: reported as settled at once
    cs.setInt("kernelChoice", 2)
    cs.dispatch()
    st = cs.stats()
    assert (st["kernel_choice"], st["kernel_choice_settled"]) == (2, 1)
    cs.Delete()


def test_a_scene_that_comes_back_skips_the_kernel_trial(scene4):
    """The trial's result is remembered per scene content (knob tuneCache): a second context that loads the same scene is
    settled from its first frame on, with the same choice."""
    from raytracing_amd import ComputeShader
    a = ComputeShader()
    setup_core(a, scene4, 1280, 720, 16, 8)
    for _ in range(8):      # one that allocates, four kernel trials, up to two for the pass shape
        a.dispatch()
    sa = a.stats()
    assert sa["kernel_choice_settled"] == 1
    b = ComputeShader()
    setup_core(b, scene4, 1280, 720, 16, 8)
    b.dispatch()
    sb = b.stats()
    assert (sb["kernel_choice"], sb["kernel_choice_settled"]) == (sa["kernel_choice"], 1)
    a.Delete()
    b.Delete()
