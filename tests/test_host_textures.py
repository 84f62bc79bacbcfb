"""CPU: the loader's texture decoders (raytracing_amd/csrc/host/png.h, jpeg.h) behind rth_decode_image, which stands
where the reference calls stbi_load (Texture.cpp:9).

PARITY UNPINNED: stb_image is not in the reference's tree (only stb_image_write / stb_truetype are), so nothing here is
compared with stb_image itself.  PNG decoding is lossless and is checked exactly against PIL's decode of the same file
plus the documented 8-bit reduction rules; JPEG decoders differ in the last bits of the inverse DCT, the chroma
upsampling filter and the colour conversion, so the JPEG checks allow 4 / 255 per sample against libjpeg-turbo (PIL) and
a small mean difference.  (Measured on the two JPEGs the reference's models carry — sphere: progressive 4:2:0, burger:
baseline 4:2:0 — max 3, mean 0.04 / 0.11; those files are not part of this repository.)"""
import struct
import zlib

import numpy as np
import pytest
from PIL import Image

from raytracing_amd import scene as S


def _picture(w, h, seed=3):
    """Smooth colour ramps with some texture: what chroma subsampling is meant for."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 9.0) * np.cos(y / 13.0), 127 + 100 * np.cos((x + y) / 17.0), 40 + 180 * (x / max(1, w - 1))], -1)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


JPEG_VARIANTS = {
    "baseline_444": dict(subsampling=0, quality=92),
    "baseline_422": dict(subsampling=1, quality=90),
    "baseline_420": dict(subsampling=2, quality=85),
    "optimised_tables": dict(subsampling=2, quality=80, optimize=True),
    "progressive_420": dict(subsampling=2, quality=88, progressive=True),
    "progressive_444": dict(subsampling=0, quality=95, progressive=True),
    "restart_markers": dict(subsampling=2, quality=85, restart_marker_blocks=3),
    "restart_rows_progressive": dict(subsampling=1, quality=85, progressive=True, restart_marker_rows=1),
    "low_quality": dict(subsampling=2, quality=25),
}


@pytest.mark.parametrize("size", [(97, 61), (16, 16), (1, 1), (33, 8), (8, 35)])
@pytest.mark.parametrize("variant", list(JPEG_VARIANTS))
def test_jpeg_close_to_libjpeg(tmp_path, variant, size):
    w, h = size
    p = tmp_path / "t.jpg"
    Image.fromarray(_picture(w, h)).save(p, "JPEG", **JPEG_VARIANTS[variant])
    got = S.decode_image(p)
    want = np.asarray(Image.open(p).convert("RGB")).astype(int)
    assert got.shape == (h, w, 4) and (got[..., 3] == 255).all()
    d = np.abs(got[..., :3].astype(int) - want)
    if JPEG_VARIANTS[variant]["subsampling"] == 1 and w > 2:
        # 4:2:2: stb_image's horizontal filter weighs the last-but-one output column the other way round (jpeg.h keeps
        # that, libjpeg does not): that one column is only required to be plausible
        col = 2 * ((w + 1) // 2) - 2
        assert d[:, col].max() <= 40
        d = np.delete(d, col, axis=1)
    assert d.max() <= 4, (variant, size, int(d.max()))
    assert d.mean() < 0.7, (variant, size, float(d.mean()))


def test_jpeg_greyscale_lands_in_red_only(tmp_path):
    """One component -> stbi_load reports 1 channel -> glTexImage2D(..., GL_RED, ...) (Texture.cpp:25-26)."""
    p = tmp_path / "g.jpg"
    Image.fromarray(_picture(70, 45)[..., 0]).save(p, "JPEG", quality=90)
    got = S.decode_image(p)
    want = np.asarray(Image.open(p).convert("L")).astype(int)
    assert np.abs(got[..., 0].astype(int) - want).max() <= 2
    assert (got[..., 1] == 0).all() and (got[..., 2] == 0).all() and (got[..., 3] == 255).all()


def test_jpeg_flat_blocks_are_exact(tmp_path):
    """A constant image has only DC coefficients: every decoder must return the same constant."""
    p = tmp_path / "c.jpg"
    Image.fromarray(np.full((24, 40, 3), (200, 200, 200), np.uint8)).save(p, "JPEG", quality=100, subsampling=0)
    got = S.decode_image(p)
    want = np.asarray(Image.open(p).convert("RGB"))
    assert np.array_equal(got[..., :3], want)


def test_jpeg_refuses_what_it_does_not_decode(tmp_path):
    p = tmp_path / "cmyk.jpg"
    Image.fromarray(_picture(20, 20)).convert("CMYK").save(p, "JPEG")
    with pytest.raises(RuntimeError, match="1- and 3-component"):
        S.decode_image(p)
    good = tmp_path / "ok.jpg"
    Image.fromarray(_picture(64, 64)).save(good, "JPEG", quality=90)
    raw = good.read_bytes()
    cut = tmp_path / "cut.jpg"
    cut.write_bytes(raw[:len(raw) // 2])          # data ends early: the rest of the picture keeps the decoder's zeros, no fault
    a = S.decode_image(cut)
    assert a.shape == (64, 64, 4)
    hdr = tmp_path / "hdr.jpg"
    hdr.write_bytes(raw[:40])
    with pytest.raises(RuntimeError):
        S.decode_image(hdr)
    junk = tmp_path / "junk.bin"
    junk.write_bytes(b"GIF89a" + bytes(64))
    with pytest.raises(RuntimeError, match="neither a PNG nor a JPEG"):
        S.decode_image(junk)


# ---- PNG --------------------------------------------------------------------------------------------------------

def _chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)


def _write_png(path, samples, depth, ctype, interlace=False, plte=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """Minimal PNG writer for the cases PIL cannot produce.  samples: [H, W, C] integers of `depth` bits."""
    H, W, C = samples.shape

    def pack_rows(img):
        h, w, _ = img.shape
        out = bytearray()
        prev = None
        for y in range(h):
            vals = img[y].reshape(-1)
            if depth == 16:
                row = vals.astype(">u2").tobytes()
            elif depth == 8:
                row = vals.astype(np.uint8).tobytes()
            else:
                bits = np.zeros(((len(vals) * depth + 7) // 8) * 8, np.uint8)
                for b in range(depth):
                    bits[np.arange(len(vals)) * depth + b] = (vals >> (depth - 1 - b)) & 1
                row = np.packbits(bits).tobytes()
            bpp = max(1, C * depth // 8)
            ft = filters[y % len(filters)]
            cur = np.frombuffer(row, np.uint8).astype(int)
            up = prev if prev is not None else np.zeros_like(cur)
            left = np.concatenate([np.zeros(bpp, int), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            ul = np.concatenate([np.zeros(bpp, int), up[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if ft == 0:
                enc = cur
            elif ft == 1:
                enc = cur - left
            elif ft == 2:
                enc = cur - up
            elif ft == 3:
                enc = cur - (left + up) // 2
            else:
                pa, pb, pc = np.abs(up - ul), np.abs(left - ul), np.abs(left + up - 2 * ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
                enc = cur - pred
            out += bytes([ft]) + (enc & 255).astype(np.uint8).tobytes()
            prev = cur
        return bytes(out)

    if not interlace:
        raw = pack_rows(samples)
    else:
        raw = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += pack_rows(sub)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        data += _chunk(b"PLTE", bytes(plte))
    if trns is not None:
        data += _chunk(b"tRNS", bytes(trns))
    z = zlib.compress(raw, 6)
    data += _chunk(b"IDAT", z[:len(z) // 2]) + _chunk(b"IDAT", z[len(z) // 2:]) + _chunk(b"IEND", b"")
    path.write_bytes(data)


def _rgba(r, g=None, b=None, a=None):
    out = np.zeros(r.shape + (4,), np.uint8)
    out[..., 0] = r
    out[..., 1] = 0 if g is None else g
    out[..., 2] = 0 if b is None else b
    out[..., 3] = 255 if a is None else a
    return out


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("size", [(37, 23), (1, 1), (5, 3), (8, 8), (3, 17)])
def test_png_every_colour_type_and_depth(tmp_path, size, interlace):
    w, h = size
    rng = np.random.default_rng(w * 100 + h)
    p = tmp_path / "t.png"
    # grey, 1 / 2 / 4 / 8 / 16 bits: spread over 0..255 (x 255, 85, 17, 1) or cut to the high byte
    for depth, mul in ((1, 255), (2, 85), (4, 17), (8, 1)):
        v = rng.integers(0, 1 << depth, (h, w, 1))
        _write_png(p, v, depth, 0, interlace)
        assert np.array_equal(S.decode_image(p), _rgba((v[..., 0] * mul).astype(np.uint8))), ("grey", depth)
    v = rng.integers(0, 1 << 16, (h, w, 1))
    _write_png(p, v, 16, 0, interlace)
    assert np.array_equal(S.decode_image(p), _rgba((v[..., 0] >> 8).astype(np.uint8)))
    # RGB and RGBA, 8 and 16 bits
    for depth in (8, 16):
        v = rng.integers(0, 1 << depth, (h, w, 3))
        _write_png(p, v, depth, 2, interlace)
        v8 = (v >> (depth - 8)).astype(np.uint8)
        assert np.array_equal(S.decode_image(p), _rgba(v8[..., 0], v8[..., 1], v8[..., 2])), ("rgb", depth)
        v = rng.integers(0, 1 << depth, (h, w, 4))
        _write_png(p, v, depth, 6, interlace)
        v8 = (v >> (depth - 8)).astype(np.uint8)
        assert np.array_equal(S.decode_image(p), _rgba(v8[..., 0], v8[..., 1], v8[..., 2], v8[..., 3])), ("rgba", depth)
    # palette, 1 / 2 / 4 / 8 bits, without and with tRNS (3 -> 4 channels)
    for depth in (1, 2, 4, 8):
        n = 1 << depth
        pal = rng.integers(0, 256, (n, 3)).astype(np.uint8)
        v = rng.integers(0, n, (h, w, 1))
        _write_png(p, v, depth, 3, interlace, plte=pal.reshape(-1))
        c = pal[v[..., 0]]
        assert np.array_equal(S.decode_image(p), _rgba(c[..., 0], c[..., 1], c[..., 2])), ("palette", depth)
        alpha = rng.integers(0, 256, max(1, n // 2)).astype(np.uint8)          # shorter than the palette: the rest is opaque
        _write_png(p, v, depth, 3, interlace, plte=pal.reshape(-1), trns=alpha)
        a = np.where(v[..., 0] < len(alpha), alpha[np.minimum(v[..., 0], len(alpha) - 1)], 255).astype(np.uint8)
        assert np.array_equal(S.decode_image(p), _rgba(c[..., 0], c[..., 1], c[..., 2], a)), ("palette+tRNS", depth)
    # a tRNS colour key on RGB: 4 channels, alpha 0 where the pixel equals the key
    v = rng.integers(0, 4, (h, w, 3)) * 60
    key = v[h // 2, w // 2]
    _write_png(p, v, 8, 2, interlace, trns=struct.pack(">HHH", *[int(k) for k in key]))
    a = np.where((v == key).all(-1), 0, 255).astype(np.uint8)
    assert np.array_equal(S.decode_image(p), _rgba(v[..., 0].astype(np.uint8), v[..., 1].astype(np.uint8), v[..., 2].astype(np.uint8), a))


def test_png_agrees_with_pil_on_files_pil_writes(tmp_path):
    pic = _picture(61, 47)
    p = tmp_path / "a.png"
    for mode, arr in (("RGB", pic), ("RGBA", np.dstack([pic, pic[..., 0]])), ("L", pic[..., 1])):
        Image.fromarray(arr, mode).save(p, optimize=True)
        got = S.decode_image(p)
        if mode == "L":
            assert np.array_equal(got, _rgba(arr))
        elif mode == "RGB":
            assert np.array_equal(got, _rgba(arr[..., 0], arr[..., 1], arr[..., 2]))
        else:
            assert np.array_equal(got, arr)
    Image.fromarray(pic).quantize(32).save(p)
    assert np.array_equal(S.decode_image(p)[..., :3], np.asarray(Image.open(p).convert("RGB")))
    assert np.array_equal(S.decode_png(p), S.decode_image(p))          # the older name of the same call


def test_png_two_channels_and_damage_are_refused(tmp_path):
    p = tmp_path / "la.png"
    Image.fromarray(np.dstack([_picture(9, 9)[..., 0], _picture(9, 9)[..., 1]]), "LA").save(p)
    with pytest.raises(RuntimeError, match="2-channel"):        # Texture.cpp:27 throws for numColCh == 2
        S.decode_image(p)
    v = np.zeros((4, 4, 1), int)
    _write_png(p, v, 8, 0, trns=struct.pack(">H", 0))            # grey + colour key = 2 channels as well
    with pytest.raises(RuntimeError, match="2-channel"):
        S.decode_image(p)
    good = tmp_path / "g.png"
    Image.fromarray(_picture(32, 32)).save(good)
    raw = bytearray(good.read_bytes())
    bad = tmp_path / "bad.png"
    bad.write_bytes(bytes(raw[:len(raw) - 30]))
    with pytest.raises(RuntimeError):
        S.decode_image(bad)
    raw[24] = 3                                                   # bit depth 3
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError, match="bit depth"):
        S.decode_image(bad)
