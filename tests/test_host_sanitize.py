"""CPU: the host layer (GLTF loader, BVH builder, PNG decoder) under AddressSanitizer + UBSan + LeakSanitizer, through
its C entry points in dry-run mode: every scene twice, failing loads followed by good ones, degenerate builder input,
truncated and corrupted PNG files (tools/sanitize/).  GPU code cannot be sanitized on this pool; this is the host half."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_is_clean_under_sanitizers(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    out = r.stdout
    if "host sanitizer run:" not in out and ("cannot find -lasan" in out or "ASan runtime does not come first" in out
                                             or "libasan" in out and "No such file" in out):
        pytest.skip("no usable AddressSanitizer runtime in this environment: " + out[-300:])
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out and "LeakSanitizer" not in out, out[-3000:]
    assert "host sanitizer run: 0 failed expectation(s)" in out, out[-3000:]
    assert r.returncode == 0, out[-3000:]
