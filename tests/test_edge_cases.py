"""Edge cases of the boundary: descriptions the library must refuse (checked before any device
call, so these run without a GPU), and on the GPU the sizes and limits the reference's render()
handles: image sizes that are not a multiple of the tile, a single pixel, no objects at all,
max_bounces 0 and 1, sample counts beyond samples_per_pixel."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from conftest import scene_path

MINI = """
camera: {type: perspective, from: [0, 0, 5], to: [0, 0, 0], vfov: 40}
render: {width: 13, height: 7, samples: 3, max_bounces: 4}
environment:
  texture: {type: solid_color, color: [0.2, 0.4, 0.8]}
objects:
  - type: sphere
    center: [0, 0, 0]
    radius: 1
    material: {type: lambertian, texture: {type: solid_color, color: [0.8, 0.3, 0.3]}}
"""
EMPTY = """
camera: {type: orthographic, from: [0, 0, 5], to: [0, 0, 0], width: 4}
render: {width: 9, height: 5, samples: 2, max_bounces: 3}
environment:
  texture: {type: solid_color, color: [0.25, 0.5, 0.75]}
objects: []
"""


def clone(desc):
    out = type(desc)()
    C.memmove(C.byref(out), C.byref(desc), C.sizeof(desc))
    return out


def create(crt, desc):
    h = C.c_void_p()
    rc = crt.lib().crt_scene_create(C.byref(desc), C.byref(h))
    return rc, crt.lib().crt_last_error().decode()


def test_invalid_descriptions_are_refused(crt):
    sc = crt.parse_scene(MINI)
    base = sc.desc
    lib = crt.lib()

    d = clone(base); d.background_index = 7
    rc, msg = create(crt, d)
    assert rc == -1 and "background_index" in msg
    d = clone(base); d.render.image_width = 0
    assert create(crt, d)[0] == -1
    d = clone(base); d.render.max_bounces = -1
    assert create(crt, d)[0] == -1
    d = clone(base); d.camera.type = 9
    assert create(crt, d)[0] == -1

    objs = (type(base.objects.contents) * 1)()
    C.memmove(objs, base.objects, C.sizeof(objs))
    objs[0].mat_idx = 5
    d = clone(base); d.objects = C.cast(objs, type(base.objects))
    rc, msg = create(crt, d)
    assert rc == -1 and "material index" in msg

    n = base.bvh.slot_count
    children = (C.c_int32 * n)(*[base.bvh.children[i] for i in range(n)])
    children[0] = 0                                              # the root as its own child
    d = clone(base); d.bvh.children = C.cast(children, type(base.bvh.children))
    rc, msg = create(crt, d)
    assert rc == -1 and "BVH" in msg
    oi = (C.c_int32 * n)(*[base.bvh.object_index[i] for i in range(n)])
    oi[0] = 99
    d = clone(base); d.bvh.object_index = C.cast(oi, type(base.bvh.object_index))
    assert create(crt, d)[0] == -1

    # options out of range (a work item count near INT_MAX would overflow the chunk arithmetic)
    for field, value in (("work_items", 2**31 - 1), ("work_items", (1 << 26) + 1), ("exact_ties", 2), ("item_order", 3), ("chip_parts", 17), ("chip_parts", -1), ("flat_cuboids", 4), ("prim_batch", 65)):
        other = crt.parse_scene(MINI)
        other.set_options(**{field: value})
        with pytest.raises(crt.CrtError) as e:
            other.upload()
        assert e.value.status == -1 and field in str(e.value), (field, str(e.value))
        other.close()
    assert lib.crt_scene_create(None, None) == -1
    assert lib.crt_render(None, None, 1, 0, None, None) == -1
    assert lib.crt_buffers_create(0, 5, C.byref(C.c_void_p())) == -1
    sc.close()


def test_scene_without_objects_parses(crt):
    sc = crt.parse_scene(EMPTY)
    d = sc.desc
    assert d.object_count == 0 and d.bvh.slot_count == 4
    assert all(d.bvh.object_index[i] == -1 and d.bvh.children[i] == -1 for i in range(4))
    sc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1])
def test_odd_sizes_and_limits_on_gpu(crt, oracle, mode):
    from cudaraytracer_amd import api

    def compare(sc, w, h, samples, bounces):
        sc.set_render(w, h, samples, bounces)
        sc.set_mode(mode)
        dp = crt.lib().crt_host_scene_desc(sc._host)
        buf = crt.create_buffer(sc)
        ref = oracle_lib.new_passes(w, h)
        for done in range(samples):
            crt.render(sc, buf, samples=1, samples_done=done)
            oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
        got = buf.download()
        for k in oracle_lib.PASSES:
            a, b = got[k], ref[k]
            assert ((a == b) | (np.isnan(a) & np.isnan(b))).all(), (w, h, samples, bounces, k)
        buf.close()

    sc = crt.parse_scene(MINI)
    for w, h in ((13, 7), (1, 1), (8, 8), (9, 1), (1, 9), (65, 33)):
        compare(sc, w, h, 3, 4)
    compare(sc, 13, 7, 2, 1)                                     # one bounce: every path scatters once and is exhausted
    # samples beyond samples_per_pixel are clamped (render.cuh:120-124)
    sc.set_render(13, 7, 3, 4)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf, samples=100)
    full = api.last_ray_count
    crt.render(sc, buf, samples=-1)
    assert api.last_ray_count == full
    # a buffer of another size is refused
    other = crt.Buffer(5, 5)
    with pytest.raises(crt.CrtError):
        crt.render(sc, other)
    other.close()
    buf.close()
    sc.close()

    empty = crt.parse_scene(EMPTY)
    empty.set_mode(mode)
    buf = crt.create_buffer(empty)
    crt.render(empty, buf)
    got = buf.download()
    assert api.last_ray_count == 9 * 5 * 2
    assert np.allclose(got["color"], [0.25, 0.5, 0.75]) and np.allclose(got["albedo"], [0.25, 0.5, 0.75])
    assert np.isnan(got["depth"]).all() and (got["opacity"] == 0).all() and (got["normal"] == 0).all()
    buf.close()
    empty.close()


@pytest.mark.gpu
def test_zero_bounces_renders_black(crt):
    """max_bounces 0: the path loop never runs; colour and albedo are 0 (render.cuh:96-97) and no
    ray is traced.  (The reference leaves depth and opacity uninitialised in that case; here they
    are 0.)"""
    sc = crt.parse_scene(MINI.replace("max_bounces: 4", "max_bounces: 0"))
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    from cudaraytracer_amd import api
    got = buf.download()
    assert api.last_ray_count == 0
    assert (got["color"] == 0).all() and (got["albedo"] == 0).all() and (got["opacity"] == 0).all()
    buf.close()
    sc.close()


@pytest.mark.gpu
def test_render_from_another_thread(crt):
    """The preview of the reference calls render() from a thread of its own (preview.cuh:294-315):
    a scene created on the main thread renders the same bits when called from a second thread,
    and two scenes render side by side from two threads."""
    import threading
    from conftest import scene_path
    sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
    sc.set_render(96, 96, 4, 8)
    buf = crt.create_buffer(sc)
    crt.render(sc, buf)
    want = buf.download()
    other = crt.load_scene(scene_path("orthographic.yaml"), base_dir=scene_path(""))
    other.set_render(64, 64, 4, 6)
    other_buf = crt.create_buffer(other)
    crt.render(other, other_buf)
    other_want = other_buf.download()

    results, errors = {}, []

    def worker(name, scene, buffer):
        try:
            for s in range(4):                                   # the preview pattern: one sample per call
                crt.render(scene, buffer, samples=1, samples_done=s)
            results[name] = buffer.download()
        except Exception as e:                                   # noqa: BLE001 - reported below
            errors.append((name, e))

    threads = [threading.Thread(target=worker, args=("cornell", sc, buf)),
               threading.Thread(target=worker, args=("ortho", other, other_buf))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for got, ref in ((results["cornell"], want), (results["ortho"], other_want)):
        for k in got:
            a, b = got[k], ref[k]
            # one call of 4 samples and 4 calls of one sample fold the mean differently (last bits)
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6, equal_nan=True), k
    for x in (buf, other_buf):
        x.close()
    sc.close()
    other.close()


@pytest.mark.gpu
def test_deterministic_option_makes_bulk_renders_bit_reproducible(crt):
    """A bulk call sums the samples of a chunk in the order their paths finish, so with several
    samples per chunk (any full-size frame) two runs agree only up to float summation order.
    crt_scene_options.deterministic renders one sample per chunk and adds the chunks in sample
    order: two runs of the full-size frame then agree in every bit, and the result equals the
    default render to that summation-order tolerance."""
    images = {}
    for deterministic in (1, 0):
        sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
        sc.set_render(1000, 1000, samples=12)
        sc.set_options(deterministic=deterministic)
        buf = crt.create_buffer(sc)
        runs = []
        for _ in range(2):
            crt.render(sc, buf)
            runs.append(buf.download())
        images[deterministic] = runs
        buf.close()
        sc.close()
    for k in oracle_lib.PASSES:
        a, b = images[1]
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
        c = images[0][0]
        np.testing.assert_allclose(c[k], a[k], rtol=2e-5, atol=1e-6, equal_nan=True, err_msg=k)
        np.testing.assert_allclose(images[0][1][k], a[k], rtol=2e-5, atol=1e-6, equal_nan=True, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,mode", [("orthographic.yaml", 0), ("materials.yaml", 0), ("obj.yaml", 1), ("features_small.yaml", 2),
                                        ("cornell_box.yaml", 0)])
def test_many_samples_of_a_pixel_in_one_wave_lose_no_addition(crt, scene, mode):
    """Tile sums are added without atomics; lanes that hold samples of the same pixel take turns
    (a tag word per pixel).  With one chunk for all samples every work item holds 48 samples of
    each of its 32 pixels, so a wave always carries several samples of a pixel.  A lost or doubled
    addition would change a pixel by about 1/48 of a sample's value; against the render with one
    sample per chunk (no two samples of a pixel ever in flight together) only the summation order
    may differ."""
    from cudaraytracer_amd import api
    images, rays = {}, {}
    for deterministic in (1, 0):
        sc = crt.load_scene(scene_path(scene), base_dir=scene_path(""))
        sc.set_render(96, 64, samples=48)
        if deterministic:
            sc.set_options(deterministic=1)
        else:
            sc.set_options(work_items=1)                 # one chunk: 48 samples per pixel and item
        sc.set_mode(mode)
        buf = crt.create_buffer(sc)
        crt.render(sc, buf)
        images[deterministic], rays[deterministic] = buf.download(), api.last_ray_count
        buf.close()
        sc.close()
    assert rays[0] == rays[1] and rays[0] > 0
    for k in oracle_lib.PASSES:
        assert np.array_equal(np.isnan(images[0][k]), np.isnan(images[1][k])), k
        np.testing.assert_allclose(images[0][k], images[1][k], rtol=3e-5, atol=2e-6, equal_nan=True, err_msg=k)
