"""Seeded random scenes: every object type, wrapper and material the scene format has, in random
numbers and places, rendered by the HIP path in each geometry mode and by the CPU oracle.  All six
passes and the ray count must be equal bit for bit, sample after sample (see the one exception,
which only very large images can show, at the comparison below).  Wrappers are nested at
random too, which includes the combinations whose boxes do not contain their geometry.

Image textures and normal maps sit on triangles and on spheres, and a quarter of the scenes have an
environment image: the texture coordinates of spheres and of the environment go through acos /
atan2 (and a sphere's tangent through sin / cos), which the kernels evaluate with the plain-float
functions of det_math.h; the oracle is switched to the same evaluations for this test."""
import os
import random

import numpy as np
import pytest

import oracle_lib

pytestmark = pytest.mark.gpu

# "ordered": the reference's tree walked in the reference's order even where the library would
# have chosen a hierarchy and an order of its own
# "exact_*": the library's own hierarchy and order with crt_scene_options.exact_ties: near-ties decided like the
# reference's ordered walk decides them, so these too must agree in every bit
MODES = {"auto": 0, "tree_hbm": 1, "tree_lds": 2, "ordered": 1, "exact_hbm": 1, "exact_lds": 2}


_SCALE = int(os.environ.get("CRT_FUZZ_SCALE", "1"))
_STRICT = os.environ.get("CRT_FUZZ_STRICT") == "1"      # long hunts: report every path that takes the other of two near-tied hits


def _vec(rng, lo, hi):
    return "[%s]" % ", ".join("%.4f" % rng.uniform(lo, hi) for _ in range(3))


def _texture(rng, depth=0):
    if depth < 2 and rng.random() < 0.3:
        return "{type: checker, scale: %.3f, odd: %s, even: %s}" % (rng.uniform(0.2, 1.5), _texture(rng, depth + 1), _texture(rng, depth + 1))
    return "{type: solid_color, color: %s}" % _vec(rng, 0.05, 0.95)


def _material(rng, flat_surface=False):
    kind = rng.random()
    if flat_surface and kind < 0.15:    # image lookups on triangles are plain arithmetic on the barycentrics
        return "{type: lambertian, texture: {type: image, filename: %s}}" % rng.choice(["rgb.png", "grey.png", "rgba.png"])
    if kind < 0.45:
        return "{type: lambertian, texture: %s}" % _texture(rng)
    if kind < 0.7:
        fuzz = rng.choice([0.0, rng.uniform(0.05, 0.9)])
        return "{type: metal, color: %s, fuzz: %.3f}" % (_vec(rng, 0.3, 1.0), fuzz)
    if kind < 0.85:
        return "{type: dielectric, ior: %.3f}" % rng.uniform(1.1, 2.2)
    return "{type: diffuse_light, texture: {type: solid_color, color: %s}}" % _vec(rng, 0.5, 6.0)


def _object(rng, extent, wrappers, depth=0):
    kind = rng.random()
    if depth < wrappers and kind < 0.2:
        inner = _object(rng, extent, wrappers, depth + 1)
        if rng.random() < 0.5:
            return "{type: translate, offset: %s, object: %s}" % (_vec(rng, -1.0, 1.0), inner)
        anchor = ", anchor: %s" % _vec(rng, -extent, extent) if rng.random() < 0.3 else ""
        return "{type: rotate, angles: %s%s, object: %s}" % (_vec(rng, -80.0, 80.0), anchor, inner)
    mat = _material(rng)
    if rng.random() < 0.04:             # degenerate shapes: no area, no volume, huge, far away
        pick = rng.randrange(5)
        p = _vec(rng, -extent, extent)
        if pick == 0:
            return "{type: tri, v0: %s, v1: %s, v2: %s, material: %s}" % (p, p, _vec(rng, -extent, extent), mat)
        if pick == 1:
            return "{type: sphere, center: %s, radius: 0, material: %s}" % (p, mat)
        if pick == 2:
            return "{type: quad, corner: %s, edge1: [1.5, 0.5, 0], edge2: [3, 1, 0], material: %s}" % (p, mat)
        if pick == 3:
            return "{type: sphere, center: [0, -1000, 0], radius: %.3f, material: %s}" % (1000 - extent - rng.uniform(0.0, 1.0), mat)
        return "{type: box, corner1: [1.0e6, 1.0e6, 1.0e6], corner2: [1.0e6, 1.0e6, 1.00001e6], material: %s}" % mat
    if kind >= 0.45 and kind < 0.93:
        mat = _material(rng, True)
        if rng.random() < 0.2:
            mat += ", normal: {type: image, filename: rgb.png}"
    if kind < 0.45:
        if rng.random() < 0.3:          # textured and normal-mapped spheres: (u, v) and the tangent through acos/atan2/sin/cos
            mat = _material(rng, True)
            if rng.random() < 0.4:
                mat += ", normal: {type: image, filename: %s}" % rng.choice(["rgb.png", "grey.png"])
        c = _vec(rng, -extent, extent)
        moving = ", center2: %s" % _vec(rng, -extent, extent) if rng.random() < 0.25 else ""
        return "{type: sphere, center: %s%s, radius: %.4f, material: %s}" % (c, moving, rng.uniform(0.1, 0.9), mat)
    if kind < 0.6:
        return "{type: tri, v0: %s, v1: %s, v2: %s, material: %s}" % (_vec(rng, -extent, extent), _vec(rng, -extent, extent), _vec(rng, -extent, extent), mat)
    if kind < 0.78:
        return "{type: quad, corner: %s, edge1: %s, edge2: %s, material: %s}" % (_vec(rng, -extent, extent), _vec(rng, -2.0, 2.0), _vec(rng, -2.0, 2.0), mat)
    if kind < 0.93:
        c1 = [rng.uniform(-extent, extent) for _ in range(3)]
        c2 = [v + rng.uniform(0.2, 1.5) for v in c1]
        return "{type: box, corner1: [%.4f, %.4f, %.4f], corner2: [%.4f, %.4f, %.4f], material: %s}" % (*c1, *c2, mat)
    smooth = "true" if rng.random() < 0.5 else "false"
    return "{type: uv_sphere, center: %s, radius: %.4f, rings: %d, segments: %d, shade_smooth: %s, material: %s}" % (
        _vec(rng, -extent, extent), rng.uniform(0.3, 1.0), rng.randint(3, 6), rng.randint(4, 9), smooth, mat)


def random_scene(seed, count, scale=None):
    scale = _SCALE if scale is None else scale
    rng = random.Random(seed)
    wrappers = 1 + seed % 2          # odd seeds nest wrappers, even seeds keep boxes and geometry together
    extent = 2.0 + 0.15 * count ** 0.5
    if rng.random() < 0.7:
        defocus = ", defocus_angle: %.3f, focus_dist: %.3f" % (rng.uniform(0.2, 2.0), rng.uniform(4.0, 12.0)) if rng.random() < 0.5 else ""
        camera = "{type: perspective, from: %s, to: %s, vfov: %.2f%s}" % (_vec(rng, 5.0, 9.0), _vec(rng, -0.5, 0.5), rng.uniform(25.0, 70.0), defocus)
    else:
        camera = "{type: orthographic, from: %s, to: %s, width: %.3f}" % (_vec(rng, 5.0, 9.0), _vec(rng, -0.5, 0.5), rng.uniform(4.0, 10.0))
    lines = ["camera: " + camera,
             "render: {width: %d, height: %d, samples: 3, max_bounces: %d}" % (scale * rng.randint(9, 70), scale * rng.randint(5, 45), rng.choice([1, 2, 3, 4, 6, 9, 30])),
             "environment:",
             ("  texture: {type: image, filename: %s}" % rng.choice(["rgb.png", "rgba.png", "grey.png"])) if rng.random() < 0.25 else
             ("  texture: {type: solid_color, color: %s}" % _vec(rng, 0.0, 1.0)),
             "objects:"]
    for _ in range(count):
        lines.append("  - " + _object(rng, extent, wrappers))
    return "\n".join(lines) + "\n"


CASES = [(seed, count) for seed, count in zip(range(101, 117), (1, 3, 6, 6, 9, 14, 20, 20, 33, 48, 75, 75, 120, 260, 700, 700))]
# CRT_FUZZ_EXTRA=n adds n more scenes (seeds CRT_FUZZ_BASE + k, default base 1000), CRT_FUZZ_SCALE=k renders
# them k times as wide and high (a longer hunt than the default run needs)
_COUNTS = (2, 2, 5, 8, 8, 12, 18, 18, 27, 40, 40, 60, 90, 90, 150, 300, 300, 500, 500)
CASES += [(int(os.environ.get("CRT_FUZZ_BASE", "1000")) + k, _COUNTS[k % len(_COUNTS)]) for k in range(int(os.environ.get("CRT_FUZZ_EXTRA", "0")))]


@pytest.fixture(scope="module")
def image_dir(tmp_path_factory):
    """Small image files the random scenes refer to: 3, 1 and 4 channels, odd sizes."""
    from PIL import Image
    d = tmp_path_factory.mktemp("fuzz_images")
    g = np.random.default_rng(7)
    Image.fromarray(g.integers(0, 256, (23, 37, 3), dtype=np.uint8), "RGB").save(d / "rgb.png")
    Image.fromarray(g.integers(0, 256, (16, 16), dtype=np.uint8), "L").save(d / "grey.png")
    Image.fromarray(g.integers(0, 256, (9, 31, 4), dtype=np.uint8), "RGBA").save(d / "rgba.png")
    return str(d)


@pytest.mark.parametrize("seed,count", CASES)
def test_random_scene_bit_exact_in_every_mode(crt, oracle, image_dir, seed, count):
    from cudaraytracer_amd import api
    text = random_scene(seed, count)
    oracle.oracle_set_pow5_mulchain(1)          # (1-cos)^5 as the device computes it: four multiplications
    oracle.oracle_set_det_math(1)               # acos/atan2/sin/cos as the device computes them
    try:
        reference = None
        failures = []
        for mode in MODES:
            sc = crt.parse_scene(text, base_dir=image_dir)
            if mode == "ordered":
                sc.set_options(accel=2)         # CRT_ACCEL_ORDERED
            if mode.startswith("exact"):
                sc.set_options(exact_ties=1)
            sc.set_mode(MODES[mode])            # uploads the scene
            assert mode != "ordered" or "order=reference" in sc.describe()
            assert seed % 2 or mode == "ordered" or "order=free" in sc.describe()
            assert not mode.startswith("exact") or "mode=tree" in sc.describe()
            dp = crt.lib().crt_host_scene_desc(sc._host)
            w, h = sc.desc.render.image_width, sc.desc.render.image_height
            if reference is None:
                reference = []
                ref = oracle_lib.new_passes(w, h)
                for done in range(3):
                    _, stats = oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
                    reference.append(({k: v.copy() for k, v in ref.items()}, stats["rays"]))
            buf = crt.create_buffer(sc)
            # The walk in the reference's order has to agree in every bit.  A walk in an order of the
            # library's own may take the OTHER of two hits whose ray parameters differ in the last
            # bits when the nearer one lies that much in front of its own box (a ray through an edge
            # that sits in a box face; the reference's own choice then depends on its visiting
            # order): at most one path in 100 000, which is none at all at the default image sizes.
            allowed = 0 if mode == "ordered" or mode.startswith("exact") or _STRICT else (3 * w * h + 50000) // 100000      # rounded: none below 16 667 pixels
            bad = np.zeros(w * h, bool)
            ray_gap = 0
            for done in range(3):
                crt.render(sc, buf, samples=1, samples_done=done)
                got = buf.download()
                want, rays = reference[done]
                ray_gap += abs(int(api.last_ray_count) - int(rays))
                for k in oracle_lib.PASSES:
                    a, b = got[k].reshape(w * h, -1), want[k].reshape(w * h, -1)
                    bad |= ~((a == b) | (np.isnan(a) & np.isnan(b))).all(axis=1)
            if not (bad.sum() <= allowed and ray_gap <= allowed * sc.desc.render.max_bounces):
                failures.append(f"seed {seed} ({sc.describe()}) {mode}: {bad.sum()} pixels differ ({np.flatnonzero(bad)[:8]}), ray counts off by {ray_gap}")
            buf.close()
            sc.close()
        assert reference is not None
        assert not failures, "\n".join(failures)          # (every mode is rendered before the verdict: a long hunt wants to know which modes agree)
    finally:
        oracle.oracle_set_pow5_mulchain(0)
        oracle.oracle_set_det_math(0)



# Scenes in which a long hunt (CRT_FUZZ_STRICT=1 CRT_FUZZ_EXTRA=900 CRT_FUZZ_SCALE=12) found ONE path that takes the
# other of two hits within rounding of each other: two through the BVH4 walk, one in the small-scene mode.
NEAR_TIES = [(1338, 300), (1372, 60), (1857, 5)]


@pytest.mark.parametrize("seed,count", NEAR_TIES)
def test_exact_ties_option_agrees_where_the_default_takes_the_other_near_tied_hit(crt, oracle, image_dir, seed, count):
    """crt_scene_options.exact_ties = 1: near-ties decided like the reference's ordered walk (bvh.cuh:398-411) decides
    them.  Every pass of every sample equal to the oracle's, no allowance; the default walk is off by one pixel here."""
    text = random_scene(seed, count, scale=12)
    oracle.oracle_set_pow5_mulchain(1)
    oracle.oracle_set_det_math(1)
    try:
        differing = {}
        reference = None
        for name, options in (("default", {}), ("exact", {"exact_ties": 1})):
            sc = crt.parse_scene(text, base_dir=image_dir)
            sc.set_options(**options)
            sc.upload()
            assert name != "exact" or "mode=tree" in sc.describe()
            dp = crt.lib().crt_host_scene_desc(sc._host)
            w, h = sc.desc.render.image_width, sc.desc.render.image_height
            if reference is None:
                reference = []
                ref = oracle_lib.new_passes(w, h)
                for done in range(3):
                    oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
                    reference.append({k: v.copy() for k, v in ref.items()})
            buf = crt.create_buffer(sc)
            bad = np.zeros(w * h, bool)
            for done in range(3):
                crt.render(sc, buf, samples=1, samples_done=done)
                got = buf.download()
                for k in oracle_lib.PASSES:
                    a, b = got[k].reshape(w * h, -1), reference[done][k].reshape(w * h, -1)
                    bad |= ~((a == b) | (np.isnan(a) & np.isnan(b))).all(axis=1)
            differing[name] = int(bad.sum())
            buf.close()
            sc.close()
        assert differing["exact"] == 0, differing
        assert differing["default"] <= 1, differing          # the known single path (kept visible: if it goes away the list above is stale)
    finally:
        oracle.oracle_set_pow5_mulchain(0)
        oracle.oracle_set_det_math(0)
