"""CPU: the host layer (GLTF loader, BVH builder, PNG decoder) under AddressSanitizer + UBSan + LeakSanitizer, through
its C entry points in dry-run mode: every scene twice, failing loads followed by good ones, degenerate builder input,
truncated and corrupted PNG files (tools/sanitize/).  GPU code cannot be sanitized on this pool; this is the host half."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _texture_files(d):
    """t0 .. tN for the harness to decode whole, cut short and with flipped bits."""
    import numpy as np
    from PIL import Image
    from tests.test_host_textures import JPEG_VARIANTS, _picture, _write_png
    i = 0
    for name in ("baseline_420", "baseline_422", "progressive_420", "restart_markers", "restart_rows_progressive", "optimised_tables"):
        Image.fromarray(_picture(67, 41)).save(d / ("t%d" % i), "JPEG", **JPEG_VARIANTS[name])
        i += 1
    Image.fromarray(_picture(40, 33)[..., 0]).save(d / ("t%d" % i), "JPEG", quality=80)
    i += 1
    rng = np.random.default_rng(5)
    _write_png(d / ("t%d" % i), rng.integers(0, 1 << 16, (19, 23, 4)), 16, 6, interlace=True)
    i += 1
    _write_png(d / ("t%d" % i), rng.integers(0, 4, (21, 13, 1)), 2, 3, interlace=True, plte=rng.integers(0, 256, 12).astype(np.uint8), trns=bytes([0, 128]))
    i += 1
    _write_png(d / ("t%d" % i), rng.integers(0, 2, (9, 70, 1)), 1, 0)


def test_host_layer_is_clean_under_sanitizers(tmp_path):
    tex = tmp_path / "tex"
    tex.mkdir()
    _texture_files(tex)
    env = dict(os.environ, TMPDIR=str(tmp_path), RT_SANITIZE_TEXTURES=str(tex))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    out = r.stdout
    if "host sanitizer run:" not in out and ("cannot find -lasan" in out or "ASan runtime does not come first" in out
                                             or "libasan" in out and "No such file" in out):
        pytest.skip("no usable AddressSanitizer runtime in this environment: " + out[-300:])
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-3000:]
    assert "host sanitizer run: 0 failed expectation(s)" in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
