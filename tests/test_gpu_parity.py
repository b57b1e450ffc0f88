"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: object indices and ray parameters of primary hits bit-exact; whole
images bit-exact per sample for scenes without transcendental functions on the path."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from conftest import scene_path

pytestmark = pytest.mark.gpu


def _desc_ptr(crt, sc):
    return crt.lib().crt_host_scene_desc(sc._host)


MODES = {"auto": 0, "tree_hbm": 1, "tree_lds": 2}


@pytest.mark.parametrize("mode", list(MODES))
@pytest.mark.parametrize("name,size", [("cornell_box.yaml", 256), ("orthographic.yaml", 256)])
def test_primary_hits_bit_exact(crt, oracle, name, size, mode):
    sc = crt.load_scene(scene_path(name))
    sc.set_render(size, size)
    sc.set_mode(MODES[mode])
    for sample in (1, 7):
        g = sc.trace_primary(sample)
        o = oracle_lib.trace_primary(oracle, _desc_ptr(crt, sc), size, size, sample)
        assert np.array_equal(g[0], o[0]), f"object ids differ at {np.flatnonzero(g[0] != o[0])[:10]}"
        assert np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32)), "ray parameter t not bit-exact"
        assert np.array_equal(g[4], o[4])
        hit = o[0] >= 0
        np.testing.assert_allclose(g[2][hit], o[2][hit], atol=1e-5)   # sphere (u,v) go through acosf/atan2f
        np.testing.assert_allclose(g[3][hit], o[3][hit], atol=1e-5)
    sc.close()


@pytest.mark.parametrize("mode", list(MODES))
def test_cornell_single_samples_bit_exact(crt, oracle, mode):
    """Cornell touches only + - * / sqrt: with the shared RNG every sample must match bit for bit."""
    size = 128
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    sc.set_render(size, size, samples=64)
    sc.set_mode(MODES[mode])
    buf = crt.create_buffer(sc)
    dp = _desc_ptr(crt, sc)
    ref = oracle_lib.new_passes(size, size)
    for done in range(0, 5):
        crt.render(sc, buf, samples=1, samples_done=done)
        got = buf.download()
        oracle_lib.render(oracle, dp, size, size, samples=1, samples_done=done, passes=ref)
        for k in oracle_lib.PASSES:
            # value equality: the sum starts at +0, so a -0 component comes back as +0
            a, b = got[k], ref[k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"pass {k} differs after sample {done + 1}: {np.flatnonzero(~same)[:8]}"
    buf.close()
    sc.close()


def test_cornell_bulk_matches_running_mean(crt, oracle):
    size, spp = 96, 32
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    sc.set_render(size, size, samples=spp)
    buf = crt.create_buffer(sc)
    secs = crt.render(sc, buf)
    got = buf.download()
    ref, stats = oracle_lib.render(oracle, _desc_ptr(crt, sc), size, size)
    from cudaraytracer_amd import api
    assert api.last_ray_count == stats["rays"]
    for k in oracle_lib.PASSES:
        assert np.array_equal(np.isnan(got[k]), np.isnan(ref[k]))
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-5, atol=1e-6, equal_nan=True)
    assert secs > 0
    buf.close()
    sc.close()


def test_orthographic_matches_published_passes(crt):
    """The HIP render against the reference's own published GPU output (statistical bound)."""
    from test_oracle_golden import check_against_published
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    check_against_published(crt, buf.download())
    buf.close()
    sc.close()


def test_cornell_full_size_ray_count_and_image(crt, oracle):
    """BASELINE size (1000x1000, 50 spp, 15 bounces): the device ray count must equal the
    oracle's exactly and every pass must agree to accumulation rounding."""
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    got = buf.download()
    from cudaraytracer_amd import api
    rays = api.last_ray_count
    ref, stats = oracle_lib.render(oracle, _desc_ptr(crt, sc), 1000, 1000)
    assert rays == stats["rays"]
    assert 5.7 < rays / 5e7 < 5.9                                    # 5.8 rays per path on the reference
    for k in oracle_lib.PASSES:
        np.testing.assert_allclose(got[k], ref[k], rtol=5e-5, atol=1e-5, equal_nan=True)
    # centre pixel looks at the tall box rotated by 15 degrees (normal and depth from a reference run)
    c = 500 * 1000 + 500
    np.testing.assert_allclose(got["normal"][c], [-0.258819, 0, -0.965926], atol=2e-3)
    assert abs(got["depth"][c] - 1112.56) < 3.0      # the recorded value is for "the centre pixel" (index convention unknown)
    buf.close()
    sc.close()


def test_split_render_equals_one_call(crt):
    """Progressive calling pattern (preview.cuh:306): N calls of one sample == one call of N."""
    size, spp = 64, 6
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    sc.set_render(size, size, samples=spp)
    a, b = crt.create_buffer(sc), crt.create_buffer(sc)
    crt.render(sc, a)
    for done in range(spp):
        crt.render(sc, b, samples=1, samples_done=done)
    ga, gb = a.download(), b.download()
    for k in oracle_lib.PASSES:
        assert np.array_equal(np.isnan(ga[k]), np.isnan(gb[k]))
        np.testing.assert_allclose(ga[k], gb[k], rtol=1e-5, atol=1e-6, equal_nan=True)
    # samples > samples_per_pixel is clamped (render.cuh:123)
    crt.render(sc, a, samples=1000)
    ga2 = a.download()
    np.testing.assert_allclose(ga2["color"], ga["color"], rtol=1e-6, equal_nan=True)
    a.close(); b.close(); sc.close()


def test_mesh_scene_bit_exact(crt, oracle):
    """obj.yaml (a 968-triangle mesh from an OBJ file plus the room): the BVH4 walk in HBM with
    near-first ordering and resumed walks must give the oracle's hits and images bit for bit."""
    size = 160
    sc = crt.load_scene(scene_path("obj.yaml"), base_dir=scene_path(""))
    sc.set_render(size, size, samples=8)
    assert sc.desc.object_count == 976
    dp = _desc_ptr(crt, sc)
    for sample in (1, 3):
        g = sc.trace_primary(sample)
        o = oracle_lib.trace_primary(oracle, dp, size, size, sample)
        assert np.array_equal(g[0], o[0]) and np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    buf = crt.create_buffer(sc)
    ref = oracle_lib.new_passes(size, size)
    for done in range(0, 3):
        crt.render(sc, buf, samples=1, samples_done=done)
        got = buf.download()
        oracle_lib.render(oracle, dp, size, size, samples=1, samples_done=done, passes=ref)
        for k in oracle_lib.PASSES:
            a, b = got[k], ref[k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"pass {k} differs after sample {done + 1}: {np.flatnonzero(~same)[:8]}"
    _, stats = oracle_lib.render(oracle, dp, size, size, samples=8)
    crt.render(sc, buf)
    from cudaraytracer_amd import api
    assert api.last_ray_count == stats["rays"]
    buf.close()
    sc.close()


def test_environment_image_scene_matches_oracle(crt, oracle):
    """hdr_env.yaml: a metal sphere under an HDR environment image (render.cuh:76-91: acos and
    atan2 of the ray direction, image texture lookup).  Sample by sample and bit for bit with the
    oracle evaluating acos/atan2 like the kernels (det_math.h); with the C library's functions
    (the oracle's default, like the reference's calls) the radiance agrees to 2e-3."""
    size = 256
    sc = crt.load_scene(scene_path("hdr_env.yaml"), base_dir=scene_path(""))
    sc.set_render(size, size, samples=16)
    dp = _desc_ptr(crt, sc)
    g = sc.trace_primary(1)
    o = oracle_lib.trace_primary(oracle, dp, size, size, 1)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    buf = crt.create_buffer(sc)
    from cudaraytracer_amd import api
    exact = oracle_lib.new_passes(size, size)
    oracle.oracle_set_det_math(1)
    try:
        for done in range(3):
            crt.render(sc, buf, samples=1, samples_done=done)
            got = buf.download()
            _, stats = oracle_lib.render(oracle, dp, size, size, samples=1, samples_done=done, passes=exact)
            assert api.last_ray_count == stats["rays"]
            for k in oracle_lib.PASSES:
                same = (got[k] == exact[k]) | (np.isnan(got[k]) & np.isnan(exact[k]))
                assert same.all(), (k, done, int((~same).sum()))
    finally:
        oracle.oracle_set_det_math(0)
    crt.render(sc, buf)
    got = buf.download()
    ref, stats = oracle_lib.render(oracle, dp, size, size, samples=16)
    assert api.last_ray_count == stats["rays"]
    for k in ("color", "albedo"):
        d = np.abs(got[k] - ref[k])
        assert d.max() <= 2e-3, (k, d.max())
        assert (d <= 1e-4).mean() >= 0.999, k
    for k in ("normal", "opacity", "emission"):                       # bulk sum/n vs the oracle's running mean
        np.testing.assert_allclose(got[k], ref[k], rtol=0, atol=2e-6, err_msg=k)
    assert np.array_equal(np.isnan(got["depth"]), np.isnan(ref["depth"]))
    assert got["color"].max() <= 1.0 + 1e-6          # the HDR image is tone-mapped to 8 bit at load
    buf.close()
    sc.close()


@pytest.mark.parametrize("name,mode", [("features_small.yaml", "auto"), ("features_small.yaml", "tree_lds"),
                                       ("features_small.yaml", "tree_hbm"), ("features_tree.yaml", "auto"),
                                       ("features_tree.yaml", "tree_hbm"), ("nested_wrappers.yaml", "auto"),
                                       ("nested_wrappers.yaml", "tree_hbm")])
def test_feature_scenes_bit_exact(crt, oracle, name, mode):
    """Defocus camera, moving sphere, glass, fuzzy metal, checker, image texture + normal map on a
    quad, rotated box, translated sphere, rotated sphere, smooth-shaded UV sphere, coloured
    environment; and a scene of wrappers inside wrappers, whose boxes do not contain their geometry
    and whose image therefore depends on the reference's walking order: every pass equal to the
    oracle's, sample by sample, in every geometry mode."""
    sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
    w, h = 96, 64
    sc.set_mode(MODES[mode])
    dp = _desc_ptr(crt, sc)
    for sample in (1, 2):
        g = sc.trace_primary(sample)
        o = oracle_lib.trace_primary(oracle, dp, w, h, sample)
        assert np.array_equal(g[0], o[0]) and np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    from cudaraytracer_amd import api
    buf = crt.create_buffer(sc)
    ref = oracle_lib.new_passes(w, h)
    rays = 0
    oracle.oracle_set_pow5_mulchain(1)          # (1-cos)^5 as the device computes it: four multiplications
    try:
        for done in range(0, 4):
            crt.render(sc, buf, samples=1, samples_done=done)
            rays += api.last_ray_count
            got = buf.download()
            _, stats = oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
            rays -= stats["rays"]
            for k in oracle_lib.PASSES:
                a, b = got[k], ref[k]
                same = (a == b) | (np.isnan(a) & np.isnan(b))
                assert same.all(), f"{name} {mode}: pass {k} differs after sample {done + 1}: {np.flatnonzero(~same)[:8]}"
    finally:
        oracle.oracle_set_pow5_mulchain(0)
    assert rays == 0
    buf.close()
    sc.close()


def test_fast_reciprocal_variant_is_bit_identical(crt):
    """Kernels for scenes with bounded divisors replace 1/x by v_rcp_f32 + one Newton step.  (a) The
    refined reciprocal equals the IEEE quotient for every float in [2^-100, 2^100]: exhaustive
    run over all 2^32 inputs.  (b) Both kernel variants give the same image bit for bit."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tools", "ubench", "rcp_exhaustive")
    assert os.path.exists(exe), "build it with make -C tools/ubench"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
    assert "reciprocal, one Newton step:  0 mismatches" in out, out
    assert "division via reciprocal (1 step): 0 mismatches" in out, out
    # tiny numerators make the correction step underflow; the kernels divide those the ordinary way (quotient() in pt_device.h)
    assert "numerators below 2^-39: 0 mismatches" not in out, out      # (the sweep does reach the failing range)
    assert "do not send through the ordinary division): 0 mismatches" in out, out
    assert "raw v_rcp_f32 (control): 0 mismatches" not in out          # the check can fail

    images = []
    for ieee_division in (0, 1):
        for name, mode in (("cornell_box.yaml", 0), ("features_tree.yaml", 1)):
            sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
            sc.set_render(128, 96, samples=6)
            sc.set_options(ieee_division=ieee_division)
            sc.set_mode(mode)
            assert ("reciprocal=ieee" in sc.describe()) == bool(ieee_division)
            buf = crt.create_buffer(sc)
            crt.render(sc, buf)
            images.append(buf.download())
            buf.close()
            sc.close()
    for fast, ieee in ((images[0], images[2]), (images[1], images[3])):
        for k in oracle_lib.PASSES:
            assert np.array_equal(fast[k].view(np.uint32), ieee[k].view(np.uint32)), k


def test_million_triangle_scene_bit_exact(crt, oracle, tmp_path):
    """mesh1m.yaml (1 002 536 objects, BVH deeper than the 11 LDS stack entries, so the walks use
    the global part of their stacks): hits and passes equal to the oracle's, sample by sample."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    from cudaraytracer_amd import api
    obj = os.path.join(str(tmp_path), "grid708.obj")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_mesh.py"), "grid", "708", obj])
    text = open(scene_path("mesh1m.yaml")).read().replace("assets/objects/grid708.obj", obj)
    sc = crt.parse_scene(text)
    assert sc.desc.object_count == 1002536
    w, h = 96, 64
    sc.set_render(w, h, samples=3)
    dp = _desc_ptr(crt, sc)
    g = sc.trace_primary(1)
    o = oracle_lib.trace_primary(oracle, dp, w, h, 1)
    assert np.array_equal(g[0], o[0]) and np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    assert (g[0] >= 8).any()                                    # some primary rays hit the mesh itself
    buf = crt.create_buffer(sc)
    ref = oracle_lib.new_passes(w, h)
    for done in range(3):
        crt.render(sc, buf, samples=1, samples_done=done)
        rays = api.last_ray_count
        got = buf.download()
        _, stats = oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
        assert rays == stats["rays"]
        for k in oracle_lib.PASSES:
            a, b = got[k], ref[k]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            assert same.all(), f"pass {k} differs after sample {done + 1}: {np.flatnonzero(~same)[:8]}"
    buf.close()
    sc.close()


TIES = """
camera: {type: perspective, from: [0, 1, 6], to: [0, 1, 0], vfov: 40}
render: {width: 64, height: 48, samples: 2, max_bounces: 4}
environment:
  texture: {type: solid_color, color: [0.6, 0.7, 0.9]}
materials:
  - {name: red, type: lambertian, texture: {type: solid_color, color: [0.9, 0.1, 0.1]}}
  - {name: green, type: lambertian, texture: {type: solid_color, color: [0.1, 0.9, 0.1]}}
  - {name: blue, type: lambertian, texture: {type: solid_color, color: [0.1, 0.1, 0.9]}}
  - {name: mirror, type: metal, color: [0.9, 0.9, 0.9]}
objects:
  - {type: quad, corner: [-2, 0, -1], edge1: [4, 0, 0], edge2: [0, 2, 0], material: red}
  - {type: quad, corner: [-2, 0, -1], edge1: [4, 0, 0], edge2: [0, 2, 0], material: green}
  - {type: quad, corner: [-2, 0, -1], edge1: [4, 0, 0], edge2: [0, 2, 0], material: blue}
  - {type: sphere, center: [0, 1, 0.5], radius: 0.6, material: red}
  - {type: sphere, center: [0, 1, 0.5], radius: 0.6, material: mirror}
  - {type: sphere, center: [0, 1, 0.5], radius: 0.6, material: green}
  - {type: quad, corner: [-3, 0, -3], edge1: [6, 0, 0], edge2: [0, 0, 6], material: blue}
  - {type: tri, v0: [-3, 0, 3], v1: [3, 0, 3], v2: [3, 0, -3], material: red}
"""


@pytest.mark.parametrize("mode", list(MODES))
def test_exact_ties_resolve_like_the_reference(crt, oracle, mode):
    """Coincident primitives give candidates with exactly equal t on every ray.  The reference's
    ordered walk keeps the LAST visited triangle (closed interval, hittable.cuh:335) and the
    FIRST visited sphere (open interval, :287-289); every geometry mode must pick the same one."""
    sc = crt.parse_scene(TIES)
    sc.set_mode(MODES[mode])
    w, h = 64, 48
    dp = _desc_ptr(crt, sc)
    for sample in (1, 2):
        g = sc.trace_primary(sample)
        o = oracle_lib.trace_primary(oracle, dp, w, h, sample)
        assert np.array_equal(g[0], o[0]), np.flatnonzero(g[0] != o[0])[:10]
        assert np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))
    assert len(set(o[0].tolist()) - {-1}) >= 3                   # wall, sphere and floor all visible
    buf = crt.create_buffer(sc)
    ref = oracle_lib.new_passes(w, h)
    oracle.oracle_set_pow5_mulchain(1)
    try:
        for done in range(2):
            crt.render(sc, buf, samples=1, samples_done=done)
            got = buf.download()
            oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
            for k in oracle_lib.PASSES:
                a, b = got[k], ref[k]
                assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (mode, k, done)
    finally:
        oracle.oracle_set_pow5_mulchain(0)
    buf.close()
    sc.close()


def test_kernel_variant_selection(crt):
    """Automatic choice of the geometry path and of the reciprocal form per scene."""
    def describe(name, **kw):
        sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
        if kw:
            sc.set_render(**kw)
        text = dict(item.split("=") for item in sc.describe().split())
        sc.close()
        return text
    d = describe("cornell_box.yaml", width=64, height=64)
    assert (d["mode"], d["slots"], d["groups"], d["reciprocal"]) == ("flat", "18", "2", "newton")
    d = describe("orthographic.yaml")
    assert d["mode"] == "flat" and d["groups"] == "0"
    d = describe("features_small.yaml")
    assert d["mode"] == "flat" and d["order"] == "free"
    d = describe("features_tree.yaml")
    assert d["mode"] in ("tree_lds", "tree_hbm") and d["objects"] == "167" and d["order"] == "free"
    # wrappers inside wrappers leave boxes and geometry apart (hittable.cuh:77-127 against :429): such
    # a scene is walked in the reference's order over the reference's tree, never by the flat mode
    d = describe("nested_wrappers.yaml")
    assert d["mode"] in ("tree_lds", "tree_hbm") and d["order"] == "reference"
    d = describe("obj.yaml", width=64, height=64)
    assert d["mode"] == "tree_hbm" and d["stack"].startswith("12+") and d["objects"] == "976"
    # a fuzz beyond the range the bound proof covers falls back to IEEE division
    sc = crt.parse_scene(TIES.replace("type: metal, color: [0.9, 0.9, 0.9]", "type: metal, color: [0.9, 0.9, 0.9], fuzz: 1.0e35"))
    assert "reciprocal=ieee" in sc.describe()
    sc.close()
