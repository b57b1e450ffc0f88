"""normal_map.yaml without its normal texture (an image-textured lambertian quad under an area
light) against the two images the reference publishes for exactly that variant:
assets/examples/normal_map/without.png (colour) and without_normal.png (normal pass).
This pins the image-texture path (JPEG decode -> RGBA8 -> software tex2D -> lambertian albedo)
end to end.  Statistical bound at the YAML's 300 spp: colour channel means within 0.3/255
(measured 0.2-0.29: 0.4 % relative; the CUDA build filters with 9-bit weights and fast-math),
RMSE <= 5 (measured 4.3); the normal pass has no Monte-Carlo noise: mean within 0.01/255,
RMSE <= 0.5 (anti-aliased edges only)."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib
from conftest import ROOT, scene_path


def golden(name):
    return np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "normal_map", name)).convert("RGB"), dtype=np.float64)


def normalize(crt, name, data):
    buf = crt.Buffer.__new__(crt.Buffer)
    buf.width, buf.height, buf.gamma, buf._h, buf._host = 1000, 1000, 2.0, None, None
    return crt.Buffer.normalize(buf, name, data).astype(np.float64)


def check(crt, passes):
    img, g = normalize(crt, "color", passes["color"]), golden("without.png")
    assert np.abs(img.reshape(-1, 3).mean(0) - g.reshape(-1, 3).mean(0)).max() <= 0.3
    assert np.sqrt(((img - g) ** 2).mean()) <= 5.0
    blocks = np.abs(img.reshape(10, 100, 10, 100, 3).mean((1, 3)) - g.reshape(10, 100, 10, 100, 3).mean((1, 3)))
    assert blocks.max() <= 1.0
    nrm, gn = normalize(crt, "normal", passes["normal"]), golden("without_normal.png")
    assert np.abs(nrm.reshape(-1, 3).mean(0) - gn.reshape(-1, 3).mean(0)).max() <= 0.01
    assert np.sqrt(((nrm - gn) ** 2).mean()) <= 0.5


def load(crt):
    crt.set_image_loader(None)
    return crt.load_scene(scene_path("normal_map_without.yaml"), base_dir=os.path.join(ROOT, "scenes"))


def test_oracle_reproduces_published_images(crt, oracle):
    sc = load(crt)
    d = sc.desc
    assert (d.object_count, d.texture_count) == (4, 3) and d.textures[0].width == 2048 and d.textures[0].channels == 4
    passes, stats = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), 1000, 1000)
    assert 1.86 < stats["rays"] / (1000 * 1000 * 300) < 1.88
    check(crt, passes)


@pytest.mark.gpu
def test_gpu_reproduces_published_images(crt):
    sc = load(crt)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    check(crt, buf.download())
    buf.close()
    sc.close()
