"""Built-in texture decoders (what stbi_load does for the reference, texture.cuh:143-158).

JPEG: the decoder restates stb_image's arithmetic (integer IDCT, centred triangle chroma
upsampling, fixed-point YCbCr).  PARITY UNPINNED at the bit level: the reference holds no decoded
pixel data for its JPEG assets and stb_image itself could not be run here, so these tests pin
the decoder against an independent one (Pillow/libjpeg-turbo, whose IDCT and upsampling round
differently: single texels may differ by a few levels) over every coding mode, and pin the
integer arithmetic against hand-computed values of the published formulas.
PNG, PNM: lossless formats, pinned exactly against Pillow."""
import io
import os

import numpy as np
import pytest

from conftest import ROOT

PIL = pytest.importorskip("PIL.Image")


def synthetic(w, h, seed=7):
    """Smooth colour gradients plus a few sharp edges and a little noise."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 23.0), 127 + 120 * np.cos(x / 17.0 - y / 7.0), 255.0 * x / max(w - 1, 1)], -1)
    img[h // 3: h // 2, w // 4: w // 2] = (250, 20, 30)
    img[: h // 5, w // 2:] = (10, 200, 240)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def save(tmp_path, name, arr, **kw):
    path = os.path.join(tmp_path, name)
    PIL.fromarray(arr).save(path, **kw)
    return path


CASES = [
    ("444", dict(subsampling=0, quality=90)),
    ("422", dict(subsampling=1, quality=85)),
    ("420", dict(subsampling=2, quality=75)),
    ("420_progressive", dict(subsampling=2, quality=80, progressive=True)),
    ("444_progressive", dict(subsampling=0, quality=95, progressive=True)),
    ("420_optimized", dict(subsampling=2, quality=60, optimize=True)),
    ("420_lowq", dict(subsampling=2, quality=15)),
]


@pytest.mark.parametrize("size", [(64, 48), (37, 29), (8, 8), (1, 1), (17, 1), (100, 3)])
@pytest.mark.parametrize("name,kw", CASES)
def test_jpeg_close_to_independent_decoder(crt, tmp_path, name, kw, size):
    w, h = size
    path = save(str(tmp_path), f"{name}_{w}x{h}.jpg", synthetic(w, h), **kw)
    mine = crt.load_image(path).astype(np.int32)
    ref = np.asarray(PIL.open(path).convert("RGB"), dtype=np.int32)
    assert mine.shape == ref.shape == (h, w, 3)
    d = np.abs(mine - ref)
    # libjpeg-turbo's merged upsampling differs from the separate triangle filter at sharp chroma
    # edges of subsampled files; everything else is IDCT and colour rounding (<= 3 levels)
    subsampled = "420" in name or "422" in name
    if not subsampled:
        assert d.mean() <= 0.6 and d.max() <= 3, (name, size, d.mean(), d.max())
    elif w * h >= 1000:
        assert d.mean() <= 1.5 and np.percentile(d, 99) <= 12, (name, size, d.mean())
    else:                                   # a handful of pixels, most of them on a chroma edge
        assert d.mean() <= 3.0, (name, size, d.mean())


def test_jpeg_grey_and_channel_requests(crt, tmp_path):
    rgb = synthetic(40, 24)
    grey = save(str(tmp_path), "grey.jpg", rgb[:, :, 1].copy(), quality=90)
    g = crt.load_image(grey)
    ref = np.asarray(PIL.open(grey), dtype=np.int32)
    assert g.shape == (24, 40, 1) and np.abs(g[:, :, 0].astype(np.int32) - ref).max() <= 2
    g4 = crt.load_image(grey, desired_channels=4)                    # grey replicated, opaque alpha
    assert g4.shape == (24, 40, 4) and (g4[:, :, 3] == 255).all()
    assert (g4[:, :, 0] == g[:, :, 0]).all() and (g4[:, :, 1] == g[:, :, 0]).all() and (g4[:, :, 2] == g[:, :, 0]).all()
    colour = save(str(tmp_path), "c.jpg", rgb, quality=90, subsampling=2)
    c3 = crt.load_image(colour)
    c4 = crt.load_image(colour, desired_channels=4)
    assert (c4[:, :, :3] == c3).all() and (c4[:, :, 3] == 255).all()
    # a one-channel request on a YCbCr file is the Y plane itself (stb_image.h:3881-3884, 4012-4014)
    y_plane = crt.load_image(colour, desired_channels=1)[:, :, 0].astype(np.int32)
    im = PIL.open(colour)
    im.draft("L", im.size)                                           # libjpeg hands out the Y plane
    assert im.mode == "L"
    assert np.abs(y_plane - np.asarray(im, dtype=np.int32)).max() <= 2
    flipped = crt.load_image(colour, flip_y=True)
    assert (flipped == c3[::-1]).all()


def test_jpeg_restart_intervals_and_cmyk(crt, tmp_path):
    rgb = synthetic(70, 50)
    for kw in (dict(restart_marker_blocks=3), dict(restart_marker_rows=1)):
        try:
            path = save(str(tmp_path), "rst.jpg", rgb, quality=85, subsampling=2, **kw)
        except TypeError:
            pytest.skip("this Pillow cannot write restart markers")
        data = open(path, "rb").read()
        if b"\xff\xdd" not in data:
            pytest.skip("this Pillow ignores the restart options")
        mine = crt.load_image(path).astype(np.int32)
        ref = np.asarray(PIL.open(path).convert("RGB"), dtype=np.int32)
        assert np.abs(mine - ref).mean() <= 1.5
    cmyk = PIL.fromarray(rgb).convert("CMYK")
    path = os.path.join(str(tmp_path), "cmyk.jpg")
    cmyk.save(path, quality=90)
    mine = crt.load_image(path).astype(np.int32)
    assert mine.shape == (50, 70, 3)
    # Adobe CMYK files store inverted inks; stb multiplies each ink by K (stb_image.h:3957-3965)
    ink = 255 - np.asarray(PIL.open(path), dtype=np.int32)
    expect = (ink[:, :, :3] * ink[:, :, 3:4] + 127) // 255
    assert np.abs(mine - expect).mean() <= 2.0


def test_jpeg_integer_formulas():
    """Hand evaluation of the published fixed-point formulas the decoder restates."""
    # YCbCr -> RGB, stb_image.h:3659-3683, for (Y, Cb, Cr) = (100, 90, 200)
    f = lambda x: int(np.float32(x) * np.float32(4096.0) + np.float32(0.5)) << 8
    y, cb, cr = 100, 90 - 128, 200 - 128
    yf = (y << 20) + (1 << 19)
    r = (yf + cr * f(1.40200)) >> 20
    g = (yf + cr * -f(0.71414) + ((cb * -f(0.34414)) & ~0xffff)) >> 20
    b = (yf + cb * f(1.77200)) >> 20
    assert (r, g, b) == (201, 62, 33)
    # triangle upsampling of a row (10, 20, 40): (3a + b + 2) >> 2 per neighbour pair
    row = [10, 20, 40]
    up = [row[0], (3 * 10 + 20 + 2) >> 2, (3 * 20 + 10 + 2) >> 2, (3 * 20 + 40 + 2) >> 2, (3 * 40 + 20 + 2) >> 2, row[2]]
    assert up == [10, 13, 18, 25, 35, 40]


def test_jpeg_known_pixels_from_handmade_file(crt, tmp_path):
    """A DC-only 4:2:0 file: every 8x8 luma block is flat, chroma is flat, so the decoded colour
    follows from the integer formulas alone: DC level d dequantised by q gives a sample
    ((d*q*4 ... ) the two-pass IDCT of a DC-only block is (d*q + 4) >> 3 + 128 after rounding."""
    img = np.zeros((16, 16, 3), np.uint8)
    img[:] = (200, 30, 60)
    path = save(str(tmp_path), "flat.jpg", img, quality=100, subsampling=2)
    mine = crt.load_image(path)
    assert mine.shape == (16, 16, 3)
    assert (mine == mine[0, 0]).all()
    assert np.abs(mine[0, 0].astype(int) - np.array([200, 30, 60])).max() <= 2


def test_jpeg_errors(crt, tmp_path):
    bad = os.path.join(str(tmp_path), "bad.jpg")
    open(bad, "wb").write(b"\xff\xd8\xff\xe0\x00\x02")
    with pytest.raises(crt.CrtError):
        crt.load_image(bad)
    with pytest.raises(crt.CrtError):
        crt.load_image(os.path.join(str(tmp_path), "missing.jpg"))


def test_scene_textures_decode(crt):
    for name in ("diff_2k.jpg", "nor_dx_2k.jpg"):
        path = os.path.join(ROOT, "scenes", "assets", "textures", "beach", name)
        mine = crt.load_image(path, desired_channels=4)
        assert mine.shape == (2048, 2048, 4) and (mine[:, :, 3] == 255).all()
        ref = np.asarray(PIL.open(path).convert("RGB"), dtype=np.int32)
        d = np.abs(mine[:, :, :3].astype(np.int32) - ref)
        assert d.max() <= 4 and d.mean() <= 0.1


def test_png_and_pnm_exact(crt, tmp_path):
    rgb = synthetic(33, 21)
    for mode, arr in (("RGB", rgb), ("L", rgb[:, :, 0].copy()), ("RGBA", np.dstack([rgb, rgb[:, :, :1]]))):
        path = save(str(tmp_path), f"{mode}.png", arr)
        mine = crt.load_image(path)
        assert (mine.reshape(arr.shape) == arr).all(), mode
    pal = PIL.fromarray(rgb).convert("P", palette=PIL.ADAPTIVE, colors=16)
    path = os.path.join(str(tmp_path), "pal.png")
    pal.save(path)
    assert (crt.load_image(path)[:, :, :3] == np.asarray(pal.convert("RGB"))).all()
    for bits in (1, 2, 4):                                           # packed grey levels and palette indices
        levels = (rgb[:, :, 0] >> (8 - bits)).astype(np.uint8)
        path = os.path.join(str(tmp_path), f"g{bits}.png")
        PIL.fromarray(levels).save(path, bits=bits) if bits != 1 else PIL.fromarray(levels * 255).convert("1").save(path)
        ref = np.asarray(PIL.open(path).convert("L"))
        got = crt.load_image(path)
        assert got.shape[:2] == ref.shape
        if got.shape[2] == 1 and bits != 1:
            pass                                                    # Pillow wrote a palette instead of grey: covered below
        assert (crt.load_image(path, desired_channels=3)[:, :, 0] == np.asarray(PIL.open(path).convert("RGB"))[:, :, 0]).all(), bits
    path = os.path.join(str(tmp_path), "i16.png")
    wide = (rgb[:, :, 0].astype(np.uint16) << 8) | 0x5a
    PIL.fromarray(wide).save(path)
    assert (crt.load_image(path)[:, :, 0] == rgb[:, :, 0]).all()     # 16-bit samples keep their high byte
    path = save(str(tmp_path), "x.ppm", rgb)
    assert (crt.load_image(path) == rgb).all()


def test_png_interlaced_and_colour_key(crt, tmp_path):
    """Adam7 and tRNS files written by hand (Pillow cannot write either)."""
    import struct
    import zlib
    rgb = synthetic(13, 11)
    w, h = 13, 11

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))

    x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
    raw = b""
    for p in range(7):
        sub = rgb[y0[p]::dy[p], x0[p]::dx[p]]
        if sub.size:
            for row in sub:
                raw += b"\x00" + row.tobytes()
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    path = os.path.join(str(tmp_path), "adam7.png")
    open(path, "wb").write(data)
    assert (np.asarray(PIL.open(path).convert("RGB")) == rgb).all()   # the hand-made file is valid
    assert (crt.load_image(path) == rgb).all()
    key = rgb[3, 4]
    raw = b"".join(b"\x00" + row.tobytes() for row in rgb)
    data = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"tRNS", struct.pack(">HHH", *[int(v) for v in key])) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    path = os.path.join(str(tmp_path), "key.png")
    open(path, "wb").write(data)
    got = crt.load_image(path)
    assert got.shape == (h, w, 4) and (got[:, :, :3] == rgb).all()
    assert ((got[:, :, 3] == 0) == (rgb == key).all(-1)).all() and got[3, 4, 3] == 0


def test_hostile_headers_are_rejected_without_allocating(crt, tmp_path):
    """Image files are untrusted input: absurd dimensions, a short IHDR and negative HDR sizes come
    back as errors through the C boundary (nothing is allocated from a header's claim, nothing is
    thrown across it), and the format is told from the file's first bytes, not from its name."""
    import struct
    import zlib
    from PIL import Image

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    sig = b"\x89PNG\r\n\x1a\n"
    huge = sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"")) + chunk(b"IEND", b"")
    short = sig + chunk(b"IHDR", struct.pack(">II", 4, 4)) + chunk(b"IEND", b"")
    cases = {"huge.png": huge, "short.png": short,
             "neg.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y -5 +X 7\n" + b"\0" * 64,
             "big.hdr": b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2000000000 +X 2000000000\n",
             "big.ppm": b"P6\n99999999999 3\n255\n" + b"\0" * 16,
             "big2.ppm": b"P6\n70000 70000\n255\n" + b"\0" * 16}
    for name, data in cases.items():
        path = tmp_path / name
        path.write_bytes(data)
        with pytest.raises(crt.CrtError):
            crt.load_image(str(path))
    # a PNG behind another name decodes (content sniffing, like stbi_load)
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    Image.fromarray(img, "RGB").save(tmp_path / "real.png")
    os.rename(tmp_path / "real.png", tmp_path / "texture.tex")
    assert np.array_equal(crt.load_image(str(tmp_path / "texture.tex")), img)
    (tmp_path / "unknown.bmp").write_bytes(b"BM" + b"\0" * 60)          # "BM" without a known header size, and no TGA either
    with pytest.raises(crt.CrtError, match="GIF, PSD and PIC are not"):
        crt.load_image(str(tmp_path / "unknown.bmp"))
    (tmp_path / "anim.gif").write_bytes(b"GIF89a" + b"\1" * 60)
    with pytest.raises(crt.CrtError, match="no built-in decoder for GIF, PSD or PIC"):
        crt.load_image(str(tmp_path / "anim.gif"))


# ---------------------------------------------------------------------------- BMP and TGA ----
# Both formats are lossless, so an independent decoder (Pillow) has to agree in every byte wherever the two
# libraries mean the same thing; stb_image's own conventions (what the reference's stbi_load returns,
# stb_image.h:5531-5732 and :5871-6074) are pinned on hand-made files.
def _bmp(width, height, bpp, rows, palette=b"", compress=0, masks=b"", header=40, top_down=False):
    """A BMP file from raw row bytes (padded here to 4 bytes)."""
    import struct
    body = b"".join(r + b"\0" * ((-len(r)) & 3) for r in rows)
    info = struct.pack("<IiiHHIIiiII", header, width, -height if top_down else height, 1, bpp, compress, len(body), 2835, 2835, 0, 0)
    info += b"\0" * (header - 40) if header in (40, 56) else b""
    offset = 14 + len(info) + len(masks) + len(palette)
    return b"BM" + struct.pack("<IHHI", offset + len(body), 0, 0, offset) + info + masks + palette + body


def test_bmp_equal_to_independent_decoder(crt, tmp_path):
    from PIL import Image
    g = np.random.default_rng(5)
    for size in ((7, 5), (8, 8), (33, 2), (1, 1)):
        rgb = g.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8)
        for mode, name in (("RGB", "rgb.bmp"), ("L", "grey.bmp"), ("P", "pal.bmp"), ("1", "mono.bmp"), ("RGBA", "rgba.bmp")):
            im = Image.fromarray(rgb, "RGB").convert(mode) if mode != "P" else Image.fromarray(rgb, "RGB").quantize(16)
            path = str(tmp_path / name)
            im.save(path)
            got = crt.load_image(path)
            want = np.asarray(Image.open(path).convert("RGBA" if mode == "RGBA" else "RGB"), dtype=np.uint8)
            assert got.shape == want.shape and np.array_equal(got, want), (size, mode)
            # channel requests go through the same conversion as for every other format
            assert crt.load_image(path, desired_channels=4).shape[2] == 4
            assert np.array_equal(crt.load_image(path, flip_y=True), got[::-1])


def test_bmp_conventions_of_stb_image(crt, tmp_path):
    import struct
    # 4 bits per pixel, odd width: high nibble first, rows bottom-up, padded to 4 bytes
    pal = bytes([10, 20, 30, 0, 40, 50, 60, 0, 70, 80, 90, 0])              # blue, green, red, unused
    (tmp_path / "p4.bmp").write_bytes(_bmp(3, 2, 4, [bytes([0x01, 0x20]), bytes([0x21, 0x00])], palette=pal))
    got = crt.load_image(str(tmp_path / "p4.bmp"))
    colours = np.array([[30, 20, 10], [60, 50, 40], [90, 80, 70]], np.uint8)
    assert np.array_equal(got, colours[np.array([[2, 1, 0], [0, 1, 2]])])    # (file's first row is the image's last)
    # the same file with a negative height is stored top-down
    (tmp_path / "p4t.bmp").write_bytes(_bmp(3, 2, 4, [bytes([0x01, 0x20]), bytes([0x21, 0x00])], palette=pal, top_down=True))
    assert np.array_equal(crt.load_image(str(tmp_path / "p4t.bmp")), got[::-1])
    # 16 bits without masks: 5-5-5, each channel widened by repeating its bits (31 -> 255, 1 -> 8, 30 -> 247)
    px = struct.pack("<HH", (31 << 10) | (1 << 5) | 30, (16 << 10) | 0 | 0)
    (tmp_path / "r555.bmp").write_bytes(_bmp(2, 1, 16, [px]))
    assert crt.load_image(str(tmp_path / "r555.bmp")).tolist() == [[[255, 8, 247], [132, 0, 0]]]
    # 16 bits with 5-6-5 masks (a 6-bit channel: 63 -> 255, 1 -> 4)
    masks = struct.pack("<III", 0xF800, 0x07E0, 0x001F)
    px = struct.pack("<HH", (31 << 11) | (1 << 5) | 0, (0 << 11) | (63 << 5) | 31)
    (tmp_path / "r565.bmp").write_bytes(_bmp(2, 1, 16, [px], compress=3, masks=masks))
    assert crt.load_image(str(tmp_path / "r565.bmp")).tolist() == [[[255, 4, 0], [0, 255, 255]]]
    # 32 bits: four channels; alpha bytes that are ALL zero mean "no alpha" (opaque), any other value is kept
    (tmp_path / "a0.bmp").write_bytes(_bmp(2, 1, 32, [bytes([1, 2, 3, 0, 4, 5, 6, 0])]))
    assert crt.load_image(str(tmp_path / "a0.bmp")).tolist() == [[[3, 2, 1, 255], [6, 5, 4, 255]]]
    (tmp_path / "a1.bmp").write_bytes(_bmp(2, 1, 32, [bytes([1, 2, 3, 0, 4, 5, 6, 9])]))
    assert crt.load_image(str(tmp_path / "a1.bmp")).tolist() == [[[3, 2, 1, 0], [6, 5, 4, 9]]]
    # run-length BMPs are refused like stb_image refuses them; so are truncated headers and absurd sizes
    for name, data in {"rle.bmp": _bmp(2, 1, 8, [b"\0\0"], palette=b"\0" * 8, compress=1),
                       "huge.bmp": _bmp(2000000000, 2000000000, 24, [b""]),
                       "masks.bmp": _bmp(2, 1, 16, [b"\0" * 4], compress=3, masks=struct.pack("<III", 7, 7, 7)),
                       "offset.bmp": _bmp(2, 1, 24, [b"\0" * 6])[:10] + struct.pack("<I", 5) + _bmp(2, 1, 24, [b"\0" * 6])[14:]}.items():
        (tmp_path / name).write_bytes(data)
        with pytest.raises(crt.CrtError):
            crt.load_image(str(tmp_path / name))


def _tga(width, height, bits, data, image_type=2, descriptor=0, colour_map=b"", map_len=0, map_bits=0, ident=b""):
    import struct
    return struct.pack("<BBBHHBHHHHBB", len(ident), 1 if colour_map else 0, image_type, 0, map_len, map_bits, 0, 0, width, height, bits, descriptor) + ident + colour_map + data


def test_tga_equal_to_independent_decoder(crt, tmp_path):
    from PIL import Image
    g = np.random.default_rng(6)
    for size in ((7, 5), (16, 3), (1, 1), (130, 2)):
        rgba = g.integers(0, 256, (size[1], size[0], 4), dtype=np.uint8)
        rgba[:, : size[0] // 2] = rgba[:, :1]                                # runs, so that the run-length form has both packet kinds
        for mode in ("RGB", "RGBA", "L", "P"):
            for rle in (False, True):
                im = Image.fromarray(rgba, "RGBA").convert(mode) if mode != "P" else Image.fromarray(rgba[:, :, :3].copy(), "RGB").quantize(32)
                path = str(tmp_path / f"{mode}_{int(rle)}.tga")
                im.save(path, compression="tga_rle" if rle else None)
                got = crt.load_image(path)
                want = np.asarray(Image.open(path).convert({"L": "L", "RGBA": "RGBA"}.get(mode, "RGB")), dtype=np.uint8)
                want = want[:, :, None] if want.ndim == 2 else want
                assert got.shape == want.shape and np.array_equal(got, want), (size, mode, rle)


def test_tga_conventions_of_stb_image(crt, tmp_path):
    import struct
    # rows are stored bottom-up unless bit 5 of the descriptor is set; colour is stored blue first
    data = bytes([1, 2, 3, 4, 5, 6])                                          # two pixels, one per row
    (tmp_path / "up.tga").write_bytes(_tga(1, 2, 24, data))
    (tmp_path / "down.tga").write_bytes(_tga(1, 2, 24, data, descriptor=0x20, ident=b"id!"))
    assert crt.load_image(str(tmp_path / "up.tga")).tolist() == [[[6, 5, 4]], [[3, 2, 1]]]
    assert crt.load_image(str(tmp_path / "down.tga")).tolist() == [[[3, 2, 1]], [[6, 5, 4]]]
    # 16-bit colour: 5-5-5 scaled as c * 255 / 31 (30 -> 246, not the 247 of bit repetition), no alpha
    px = struct.pack("<HH", (31 << 10) | (1 << 5) | 30, 0x8000 | (16 << 10))
    (tmp_path / "c16.tga").write_bytes(_tga(2, 1, 16, px, descriptor=0x20))
    assert crt.load_image(str(tmp_path / "c16.tga")).tolist() == [[[255, 8, 246], [131, 0, 0]]]
    # 16-bit grey is two channels (grey, alpha): the scene loader refuses those like the reference does (texture.cuh:152-154)
    (tmp_path / "g16.tga").write_bytes(_tga(2, 1, 16, bytes([7, 255, 9, 128]), image_type=3, descriptor=0x20))
    assert crt.load_image(str(tmp_path / "g16.tga")).tolist() == [[[7, 255], [9, 128]]]
    # colour-mapped with 16-bit entries, an index beyond the map reads entry 0; run-length packets may cross rows
    cmap = struct.pack("<HH", 31 << 10, 31 << 5)
    (tmp_path / "map.tga").write_bytes(_tga(2, 2, 8, bytes([0x82, 1, 0x00, 7]), image_type=9, descriptor=0x20, colour_map=cmap, map_len=2, map_bits=16))
    assert crt.load_image(str(tmp_path / "map.tga")).tolist() == [[[0, 255, 0], [0, 255, 0]], [[0, 255, 0], [255, 0, 0]]]
    # a file that ends early reads as zeros from there on
    (tmp_path / "short.tga").write_bytes(_tga(2, 2, 8, bytes([5, 6]), image_type=3, descriptor=0x20))
    assert crt.load_image(str(tmp_path / "short.tga"))[:, :, 0].tolist() == [[5, 6], [0, 0]]
    # not a TGA at all
    (tmp_path / "junk.tga").write_bytes(bytes(range(40)))
    with pytest.raises(crt.CrtError, match="no built-in decoder"):
        crt.load_image(str(tmp_path / "junk.tga"))


def test_bmp_texture_in_a_scene(crt, tmp_path):
    """An image texture in one of the formats a hand-made scene plausibly uses loads through the scene loader."""
    from PIL import Image
    img = (np.arange(4 * 4 * 3) * 5 % 256).astype(np.uint8).reshape(4, 4, 3)
    Image.fromarray(img, "RGB").save(tmp_path / "tex.bmp")
    Image.fromarray(img, "RGB").save(tmp_path / "tex.tga")
    for name in ("tex.bmp", "tex.tga"):
        sc = crt.parse_scene(f"""
camera: {{type: perspective, from: [0, 0, 5], to: [0, 0, 0], vfov: 40}}
render: {{width: 8, height: 8, samples: 1, max_bounces: 2}}
environment: {{texture: {{type: solid_color, color: [0, 0, 0]}}}}
objects:
  - {{type: quad, corner: [-1, -1, 0], edge1: [2, 0, 0], edge2: [0, 2, 0], material: {{type: lambertian, texture: {{type: image, filename: {name}}}}}}}
""", base_dir=str(tmp_path))
        tex = [t for t in (sc.desc.textures[i] for i in range(sc.desc.texture_count)) if t.type == 2]
        assert len(tex) == 1 and (tex[0].width, tex[0].height, tex[0].channels) == (4, 4, 4)       # three channels padded to four (texture.cuh:150-151)
        px = np.ctypeslib.as_array(tex[0].pixels, shape=(4, 4, 4))
        assert np.array_equal(px[:, :, :3], img) and (px[:, :, 3] == 255).all()
        sc.close()
