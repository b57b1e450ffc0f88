"""materials.yaml (all material types, checker, image texture, normal map) against the raw render
the reference publishes (assets/examples/materials_raw.png: 2000x400, 100 spp, 30 bounces).
Statistical bound: channel means within 0.1/255, RMSE <= 10 (noise of two independent renders
is 8.1).  The JPEG textures are decoded by the library's own decoder, which restates stb_image's
arithmetic (see test_image_decode.py for how far that is pinned)."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib
from conftest import ROOT, scene_path


def load_materials(crt):
    crt.set_image_loader(None)
    return crt.load_scene(scene_path("materials.yaml"), base_dir=os.path.join(ROOT, "scenes"))


def golden():
    return np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "materials", "materials_raw.png")).convert("RGB"), dtype=np.float64)


def check_color_against_published(crt, color):
    buf = crt.Buffer.__new__(crt.Buffer)
    buf.width, buf.height, buf.gamma, buf._h, buf._host = 2000, 400, 2.0, None, None
    img = crt.Buffer.normalize(buf, "color", color).astype(np.float64)
    g = golden()
    assert np.abs(img.reshape(-1, 3).mean(0) - g.reshape(-1, 3).mean(0)).max() <= 0.1
    assert np.sqrt(((img - g) ** 2).mean()) <= 10.0
    for k in range(6):                      # one column strip per sphere
        sl = slice(k * 333, (k + 1) * 333)
        assert abs(img[:, sl].mean() - g[:, sl].mean()) <= 0.15, k


def test_scene_indices(crt):
    sc = load_materials(crt)
    d = sc.desc
    assert (d.object_count, d.material_count, d.texture_count, d.background_index) == (16, 9, 8, 7)
    assert d.textures[4].type == 2 and d.textures[4].channels == 4 and d.textures[4].width == 2048
    assert d.objects[3].normal_idx == 5 and d.objects[3].mat_idx == 4
    assert d.materials[7].fuzz == 0.5 and d.materials[5].ior == 1.5


def test_oracle_reproduces_published_materials_render(crt, oracle):
    sc = load_materials(crt)
    passes, stats = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), 2000, 400)
    assert 3.50 < stats["rays"] / (2000 * 400 * 100) < 3.55       # 3.53 rays per path measured on the reference
    check_color_against_published(crt, passes["color"])


@pytest.mark.gpu
def test_gpu_matches_published_materials_render(crt):
    sc = load_materials(crt)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    check_color_against_published(crt, buf.download()["color"])
    buf.close()
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_vs_oracle_small(crt, oracle, mode):
    """Per-sample agreement with the oracle at reduced size.  With the Schlick power as a
    multiplication chain on both sides the only arithmetic that differs is acosf/atan2f/sinf/cosf
    (sphere uv and tangent), which feed the image-textured, normal-mapped sphere only."""
    sc = load_materials(crt)
    w, h, spp = 500, 100, 8
    sc.set_render(w, h, samples=spp)
    sc.set_mode(mode)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    got = buf.download()
    from cudaraytracer_amd import api
    rays = api.last_ray_count
    oracle.oracle_set_pow5_mulchain(1)
    try:
        ref, stats = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), w, h)
    finally:
        oracle.oracle_set_pow5_mulchain(0)
    assert abs(rays - stats["rays"]) <= 2e-4 * stats["rays"]
    for k in ("albedo", "emission", "normal", "depth", "opacity", "color"):
        a, b = got[k], ref[k]
        close = np.isclose(a, b, rtol=1e-4, atol=1e-5, equal_nan=True)
        assert close.mean() > 0.97, (k, close.mean())
        m = ~(np.isnan(a) | np.isnan(b))
        assert abs(a[m].mean() - b[m].mean()) <= 0.01 * abs(b[m].mean()) + 1e-6, k
    buf.close()
    sc.close()


@pytest.mark.gpu
def test_gpu_vs_oracle_materials_with_hdr_environment(crt, oracle):
    """materials_hdr.yaml = materials.yaml under an HDR environment image (the configuration
    "materials.yaml ... with HDR env texture"): besides the sphere uv/tangent, every escaping path
    goes through acosf/atan2f and an image fetch (render.cuh:76-91).  Same per-sample bound as the
    black-environment scene; the path structure (ray count) does not depend on the environment's
    colour, and the environment must show."""
    crt.set_image_loader(None)
    w, h, spp = 500, 100, 8
    images, rays = {}, {}
    from cudaraytracer_amd import api
    for name in ("materials.yaml", "materials_hdr.yaml"):
        sc = crt.load_scene(scene_path(name), base_dir=os.path.join(ROOT, "scenes"))
        sc.set_render(w, h, samples=spp)
        assert (sc.desc.textures[sc.desc.background_index].type == 2) == (name == "materials_hdr.yaml")
        buf = crt.create_buffer(sc)
        crt.render(sc, buf)
        images[name], rays[name] = buf.download(), api.last_ray_count
        if name == "materials_hdr.yaml":
            oracle.oracle_set_pow5_mulchain(1)
            try:
                ref, stats = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), w, h)
            finally:
                oracle.oracle_set_pow5_mulchain(0)
        buf.close()
        sc.close()
    got = images["materials_hdr.yaml"]
    assert abs(rays["materials_hdr.yaml"] - stats["rays"]) <= 2e-4 * stats["rays"]
    assert rays["materials_hdr.yaml"] == rays["materials.yaml"]
    for k in ("albedo", "emission", "normal", "depth", "opacity", "color"):
        a, b = got[k], ref[k]
        close = np.isclose(a, b, rtol=1e-4, atol=1e-5, equal_nan=True)
        assert close.mean() > 0.97, (k, close.mean())
        m = ~(np.isnan(a) | np.isnan(b))
        assert abs(a[m].mean() - b[m].mean()) <= 0.01 * abs(b[m].mean()) + 1e-6, k
    # the sky is seen directly around the box and in the metal spheres: brighter than the black variant
    assert got["color"].mean() > images["materials.yaml"]["color"].mean() + 0.01
    assert np.array_equal(np.isnan(got["depth"]), np.isnan(images["materials.yaml"]["depth"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["materials.yaml", "materials_hdr.yaml"])
@pytest.mark.parametrize("mode", [0, 1])
def test_gpu_bit_exact_with_shared_math(crt, oracle, name, mode):
    """Every material, texture kind, the normal-mapped sphere and (materials_hdr) the environment image,
    sample by sample and bit for bit: the kernels evaluate acos/atan2/sin/cos with the plain-float
    functions of det_math.h and (1-cos)^5 as a multiplication chain, and the oracle is switched to the
    same evaluations (its defaults are the C library's, like the reference's calls)."""
    crt.set_image_loader(None)
    sc = crt.load_scene(scene_path(name), base_dir=os.path.join(ROOT, "scenes"))
    w, h = 400, 80
    sc.set_render(w, h, samples=16)
    sc.set_mode(mode)
    buf = crt.create_buffer(sc)
    dp = crt.lib().crt_host_scene_desc(sc._host)
    ref = oracle_lib.new_passes(w, h)
    from cudaraytracer_amd import api
    oracle.oracle_set_pow5_mulchain(1)
    oracle.oracle_set_det_math(1)
    try:
        for done in range(3):
            crt.render(sc, buf, samples=1, samples_done=done)
            got = buf.download()
            _, stats = oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
            assert api.last_ray_count == stats["rays"], (done, api.last_ray_count, stats["rays"])
            for k in oracle_lib.PASSES:
                a, b = got[k], ref[k]
                same = (a == b) | (np.isnan(a) & np.isnan(b))
                assert same.all(), f"{name} mode {mode}: pass {k} differs after sample {done + 1} in {int((~same).sum())} values, first at {np.flatnonzero(~same)[:6]}"
    finally:
        oracle.oracle_set_pow5_mulchain(0)
        oracle.oracle_set_det_math(0)
    buf.close()
    sc.close()
