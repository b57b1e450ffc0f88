"""Pass normalisation and image writers (buffer.cuh:202-318, vec3.cuh:491-493, image.cuh:100-118)."""
import numpy as np
import pytest
from PIL import Image


def make_buf(crt, w, h, gamma=2.0):
    b = crt.Buffer.__new__(crt.Buffer)
    b.width, b.height, b.gamma, b._h, b._host = w, h, gamma, None, None
    return b


def color_value(x):
    x = np.asarray(x, np.float32)
    c = np.clip(x, np.float32(0), np.float32(0.9999999))
    return np.where(np.isnan(x), 0, (np.float32(256) * c).astype(np.int32)).astype(np.uint8)


def test_normalize_color_normal_opacity(crt):
    rng = np.random.default_rng(0)
    w, h = 17, 9
    b = make_buf(crt, w, h, gamma=2.2)
    col = (rng.random((w * h, 3)) * 1.5).astype(np.float32)
    col[3] = [0.0, 1.0, 4.0]
    got = b.normalize("color", col)
    want = color_value(np.power(col, np.float32(1 / np.float32(2.2)), dtype=np.float32)).reshape(h, w, 3)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1      # powf vs numpy power: last-ulp ties
    assert (got != want).mean() < 0.01
    nrm = (rng.random((w * h, 3)) * 2 - 1).astype(np.float32)
    assert np.array_equal(b.normalize("normal", nrm), color_value((nrm + np.float32(1)) / np.float32(2)).reshape(h, w, 3))
    op = rng.random(w * h).astype(np.float32)
    assert np.array_equal(b.normalize("opacity", op)[..., 1], color_value(op).reshape(h, w))


def test_normalize_depth_maps_nan_to_white(crt):
    w, h = 4, 2
    b = make_buf(crt, w, h)
    d = np.array([1, 2, 3, np.nan, 5, 2, 1, np.nan], np.float32)
    got = b.normalize("depth", d)[..., 0].ravel()
    want = color_value((np.where(np.isnan(d), 5, d) - 1) / np.float32(4))
    assert np.array_equal(got, want)
    assert got[3] == 255 and got[0] == 0


@pytest.mark.parametrize("fmt", ["png", "bmp", "tga", "ppm"])
def test_writers_round_trip(crt, tmp_path, fmt):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(13, 21, 3), dtype=np.uint8)
    path = str(tmp_path / f"x.{fmt}")
    rc = crt.lib().crt_write_image(path.encode(), fmt.encode(), 21, 13, img.ctypes.data)
    assert rc == 0
    back = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(back, img)


def test_builtin_png_decoder_feeds_image_textures(crt, tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(8, 16, 3), dtype=np.uint8)
    Image.fromarray(img).save(tmp_path / "t.png")
    sc = crt.parse_scene(f"""
camera: {{type: perspective, from: [0, 0, 5], to: [0, 0, 0], vfov: 40}}
render: {{width: 8, height: 8, samples: 1}}
environment:
  texture: {{type: image, filename: {tmp_path / 't.png'}, flip_y: true}}
objects: []
""")
    t = sc.desc.textures[0]
    assert (t.type, t.width, t.height, t.channels) == (2, 16, 8, 4)
    px = np.ctypeslib.as_array(t.pixels, shape=(8, 16, 4))
    assert np.array_equal(px[..., :3], img[::-1]) and np.all(px[..., 3] == 255)


def test_unsupported_format_is_an_error(crt, tmp_path):
    img = np.zeros((2, 2, 3), np.uint8)
    rc = crt.lib().crt_write_image(str(tmp_path / "x.gif").encode(), b"gif", 2, 2, img.ctypes.data)
    assert rc < 0 and b"Unsupported image format" in crt.lib().crt_last_error()


def test_jpg_writer_round_trip_and_standard_tables(crt, tmp_path):
    """-f jpg (write_jpg, image.cuh:42-44: quality 100).  The file must decode, with an
    independent decoder and with our own, to the source pixels within the format's rounding,
    and carry the typical Huffman tables of T.81 Annex K (compared with the copy Pillow writes)."""
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:37, 0:53]
    img = np.stack([xx * 4 % 256, yy * 6 % 256, (xx + yy) * 3 % 256], -1).astype(np.uint8)
    img[10:20, 10:30] = rng.integers(0, 256, (10, 20, 3), dtype=np.uint8)
    img = np.ascontiguousarray(img)
    path = str(tmp_path / "x.jpg")
    assert crt.lib().crt_write_image(path.encode(), b"jpg", 53, 37, img.ctypes.data) == 0
    back = np.asarray(Image.open(path).convert("RGB"), dtype=np.int32)
    assert back.shape == img.shape
    assert np.abs(back - img).max() <= 3 and np.abs(back - img).mean() <= 0.7
    own = crt.load_image(path).astype(np.int32)
    assert np.abs(own - img).max() <= 3

    def dht_segments(data):
        out, pos = {}, 2
        while pos + 4 <= len(data) and data[pos] == 0xFF and data[pos + 1] != 0xDA:
            length = (data[pos + 2] << 8) | data[pos + 3]
            if data[pos + 1] == 0xC4:
                body, k = data[pos + 4: pos + 2 + length], 0
                while k < len(body):
                    n = sum(body[k + 1: k + 17])
                    out[body[k]] = bytes(body[k + 1: k + 17 + n])
                    k += 17 + n
            pos += 2 + length
        return out

    Image.fromarray(img).save(str(tmp_path / "pil.jpg"), quality=90, optimize=False)
    ours, pils = dht_segments(open(path, "rb").read()), dht_segments(open(str(tmp_path / "pil.jpg"), "rb").read())
    assert set(ours) == {0x00, 0x10, 0x01, 0x11} and ours == pils
