"""Known-answer tests per building block: seeded cases through the device functions the render
kernels call (crt_probe) against the oracle's single-function probes.
  slab test     aabb::hit            aabb.cuh:132-156
  object test   hittable::hit        hittable.cuh:426-446 (+ hit_tri :314-359, hit_sphere :272-301,
                                     apply_normal_map :367-386)
  texture       _texture::value      texture.cuh:252-324
  camera ray    camera::get_ray      camera.cuh:240-278
Bit-exact everywhere (the oracle evaluates acos/atan2/sin/cos like the kernels for these tests); sphere (u, v), normal-mapped sphere
normals and image lookups through them get a tolerance that is written at the assertion."""
import ctypes as C

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

BOX, OBJECT, TEXTURE, CAMERA = 0, 1, 2, 3
N = 1000


def _f(x):
    return np.ascontiguousarray(x, np.float32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _bits_unsigned_zero(a):
    """Bit patterns with -0 read as +0.  The device skips the rotation of an object whose transform
    is the identity; the reference multiplies by the identity matrix, which turns a -0 component of
    a flipped normal into +0 (-0 + 0 = +0).  The two are equal as numbers and nothing downstream
    tells them apart (no division by a normal component, sums skip zeros)."""
    return _bits(np.ascontiguousarray(a, np.float32) + np.float32(0.0))


def _as_float(i):
    return np.asarray(i, np.int32).view(np.float32)


@pytest.fixture(scope="module")
def feature_scene(crt):
    sc = crt.load_scene(scene_path("features_small.yaml"), base_dir=scene_path(""))
    sc.set_render(96, 64, samples=4)
    yield sc
    sc.close()


def test_slab_test_known_answers(crt, oracle, feature_scene):
    g = np.random.default_rng(101)
    lo = _f(g.uniform(-5, 5, (N, 3)))
    hi = _f(lo + g.uniform(0.002, 4, (N, 3)))             # stored boxes are already padded to >= 0.001 per axis
    lo[:20], hi[:20] = np.inf, -np.inf                       # empty BVH slots (bvh.cuh:236-239)
    o = _f(g.uniform(-8, 8, (N, 3)))
    d = _f(g.normal(0, 1, (N, 3)))
    d[100:250, 0] = 0.0                                      # axis-parallel rays: the 1e8 substitute (aabb.cuh:140)
    d[200:350, 1] = 0.0
    d[300:400, 2] = _f(g.choice([1e-13, -1e-13, 1e-12, -2e-12], 100))
    o[400:500] = _f(lo[400:500] + (hi[400:500] - lo[400:500]) * g.uniform(0, 1, (100, 3)))   # origin inside the box
    o[500:560, 0] = lo[500:560, 0]                           # origin on a face, direction within it
    d[500:560, 0] = 0.0
    aim = N - 600                                            # the rest aims at the box so that many cases hit
    target = lo[600:] + (hi[600:] - lo[600:]) * g.uniform(-0.2, 1.2, (aim, 3))
    d[600:] = _f(target - o[600:])
    tmin = np.full(N, 0.001, np.float32)
    tmax = _f(np.where(g.random(N) < 0.5, np.inf, g.uniform(0.01, 3.0, N)))
    cases = np.concatenate([lo, hi, o, d, tmin[:, None], tmax[:, None]], axis=1)
    got = feature_scene.probe(BOX, cases).view(np.int32)[:, 0]
    want = np.array([oracle.oracle_aabb_hit(_ptr(lo[k]), _ptr(hi[k]), _ptr(o[k]), _ptr(d[k]), float(tmin[k]), float(tmax[k])) for k in range(N)])
    want_open = np.array([oracle.oracle_aabb_hit(_ptr(lo[k]), _ptr(hi[k]), _ptr(o[k]), _ptr(d[k]), float(tmin[k]), float("inf")) for k in range(N)])
    assert 0.2 < want.mean() < 0.8 and not want[:20].any()
    assert np.array_equal(got & 1, want), np.flatnonzero((got & 1) != want)[:10]
    # the flat mode's form picks near and far plane with fmin/fmax, which presumes min <= max: it only
    # ever sees the boxes of leaf slots, never an empty slot
    flat = (got[20:] >> 1) & 1
    assert np.array_equal(flat, want_open[20:]), 20 + np.flatnonzero(flat != want_open[20:])[:10]


def test_object_test_known_answers(crt, oracle, feature_scene):
    sc = feature_scene
    d = sc.desc
    dp = crt.lib().crt_host_scene_desc(sc._host)
    g = np.random.default_rng(202)
    n_obj = d.object_count
    obj = g.integers(0, n_obj, N)
    centre = np.array([[0.5 * (d.objects[i].bbox_min[a] + d.objects[i].bbox_max[a]) for a in range(3)] for i in obj])
    extent = np.array([[d.objects[i].bbox_max[a] - d.objects[i].bbox_min[a] for a in range(3)] for i in obj])
    o = _f(centre + g.normal(0, 1, (N, 3)) * (2.0 + 3.0 * extent.max(axis=1, keepdims=True)))
    o[:100] = _f(centre[:100])                               # from inside (back faces, far sphere root)
    target = centre + g.uniform(-0.3, 0.3, (N, 3)) * extent
    direction = _f((target - o) * g.uniform(0.3, 3.0, (N, 1)))     # unnormalised, like camera rays
    direction[:100] = _f(g.normal(0, 1, (100, 3)))
    time = _f(g.uniform(0, 1, N))
    tmin = np.full(N, 0.001, np.float32)
    tmax = _f(np.where(g.random(N) < 0.7, np.inf, g.uniform(0.2, 1.5, N)))
    cases = np.concatenate([_as_float(obj)[:, None], o, direction, time[:, None], tmin[:, None], tmax[:, None]], axis=1)
    got = sc.probe(OBJECT, cases)
    want = np.zeros((N, 11), np.float32)
    oracle.oracle_set_det_math(1)
    try:
        for k in range(N):
            out = np.zeros(10, np.float32)
            want[k, 0] = oracle.oracle_object_hit(dp, int(obj[k]), _ptr(o[k]), _ptr(direction[k]), float(time[k]), float(tmin[k]), float(tmax[k]), _ptr(out))
            if want[k, 0]:
                want[k, 1:] = out
    finally:
        oracle.oracle_set_det_math(0)
    hit = want[:, 0] > 0
    assert 0.3 < hit.mean() < 0.95
    assert np.array_equal(got[:, 0], want[:, 0]), np.flatnonzero(got[:, 0] != want[:, 0])[:10]
    is_tri = np.array([d.objects[i].type == 2 for i in obj])
    mapped = np.array([d.objects[i].normal_idx >= 0 for i in obj])
    assert (hit & is_tri).sum() > 100 and (hit & ~is_tri).sum() > 100 and (hit & mapped).sum() > 20
    # t, front face and hit point: every primitive, every transform, bit for bit
    for col in (1, 4, 8, 9, 10):
        assert np.array_equal(_bits(got[hit, col]), _bits(want[hit, col])), col
    # triangles: (u, v) interpolation and the normal (flat, smooth, rotated, normal-mapped) bit for bit
    sel = hit & is_tri
    assert np.array_equal(_bits(got[sel][:, 2:4]), _bits(want[sel][:, 2:4]))
    assert np.array_equal(_bits_unsigned_zero(got[sel][:, 5:8]), _bits_unsigned_zero(want[sel][:, 5:8]))
    # spheres: outward normal bit for bit without a normal map
    sel = hit & ~is_tri & ~mapped
    assert np.array_equal(_bits_unsigned_zero(got[sel][:, 5:8]), _bits_unsigned_zero(want[sel][:, 5:8]))
    # sphere (u, v) through acos / atan2, and the normal of a normal-mapped sphere through the tangent frame
    # (sin / cos of that angle) and a texel looked up at (u, v): the kernels use the plain-float functions of
    # det_math.h and the oracle was switched to them above, so these agree bit for bit as well
    sel = hit & ~is_tri
    assert np.array_equal(_bits(got[sel][:, 2:4]), _bits(want[sel][:, 2:4]))
    sel = hit & ~is_tri & mapped
    if sel.any():                                            # (the materials scene test has such a sphere for certain)
        assert np.array_equal(_bits_unsigned_zero(got[sel][:, 5:8]), _bits_unsigned_zero(want[sel][:, 5:8]))


def test_texture_known_answers(crt, oracle, feature_scene):
    sc = feature_scene
    d = sc.desc
    dp = crt.lib().crt_host_scene_desc(sc._host)
    g = np.random.default_rng(303)
    tex = g.integers(0, d.texture_count, N)
    u = _f(g.uniform(-0.5, 1.5, N))                          # image lookups clamp to [0, 1] (texture.cuh:290-291)
    v = _f(g.uniform(-0.5, 1.5, N))
    p = _f(g.uniform(-6, 6, (N, 3)))
    p[:100] = np.round(p[:100])                              # checker cell borders (floor of exact integers, negatives)
    cases = np.concatenate([_as_float(tex)[:, None], u[:, None], v[:, None], p], axis=1)
    got = sc.probe(TEXTURE, cases)
    want = np.zeros((N, 3), np.float32)
    for k in range(N):
        oracle.oracle_texture_value(dp, int(tex[k]), float(u[k]), float(v[k]), _ptr(p[k]), _ptr(want[k]))
    kinds = {d.textures[int(t)].type for t in tex}
    assert kinds == {0, 1, 2}                                # solid colour, checker, image all drawn
    # solid colours, checker parity, and the software bilinear fetch (same float expression order)
    assert np.array_equal(_bits(got), _bits(want)), np.flatnonzero((_bits(got) != _bits(want)).any(axis=1))[:10]


@pytest.mark.parametrize("name", ["cornell_box.yaml", "orthographic.yaml", "features_small.yaml"])   # pinhole, orthographic, defocus + motion
def test_camera_ray_known_answers(crt, oracle, name):
    sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
    sc.set_render(200, 120, samples=8)
    dp = crt.lib().crt_host_scene_desc(sc._host)
    g = np.random.default_rng(404)
    i, j, s = g.integers(0, 200, N), g.integers(0, 120, N), g.integers(1, 9, N)
    cases = np.stack([_as_float(i), _as_float(j), _as_float(s)], axis=1)
    got = sc.probe(CAMERA, cases)
    want = np.zeros((N, 7), np.float32)
    for k in range(N):
        oracle.oracle_camera_ray(dp, int(i[k]), int(j[k]), int(s[k]), _ptr(want[k]))
    assert np.array_equal(_bits(got), _bits(want))
    sc.close()
