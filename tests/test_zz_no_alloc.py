"""crt_render_static (the entry point that allocates nothing: chunk sums and counters in a fixed
array inside the library) against crt_render, through the C ABI."""
import numpy as np
import pytest

import oracle_lib
from conftest import scene_path

pytestmark = pytest.mark.gpu


def _render_both(crt, name, width, height, spp, mode=0):
    from cudaraytracer_amd import api
    out = []
    for fn in (crt.render, crt.render_static):
        sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
        sc.set_render(width, height, samples=spp)
        sc.set_mode(mode)
        buf = crt.create_buffer(sc)
        fn(sc, buf)
        out.append((buf.download(), api.last_ray_count))
        buf.close()
        sc.close()
    return out


def _assert_same_bits(a, b):
    for k in oracle_lib.PASSES:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def test_static_scratch_entry_point_matches_crt_render(crt):
    # two calls in a row on the same GPU, the second with a smaller input: what the first call left
    # in the array must not show.  Both sizes give one sample per chunk in either entry point, so
    # the sums are added in the same order and the images agree in every bit.
    (ref, ref_rays), (got, got_rays) = _render_both(crt, "cornell_box.yaml", 160, 120, 6)
    assert got_rays == ref_rays and ref_rays > 0
    _assert_same_bits(ref, got)
    (ref, ref_rays), (got, got_rays) = _render_both(crt, "orthographic.yaml", 64, 48, 3)
    assert got_rays == ref_rays and ref_rays > 0
    _assert_same_bits(ref, got)
    # tree walk (8x4-pixel tiles, another padded size) after the flat scenes
    (ref, ref_rays), (got, got_rays) = _render_both(crt, "obj.yaml", 72, 40, 2, mode=1)
    assert got_rays == ref_rays and ref_rays > 0
    _assert_same_bits(ref, got)


def test_static_scratch_progressive_calls_fold_like_crt_render(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
    sc.set_render(96, 64, samples=4)
    a, b = crt.create_buffer(sc), crt.create_buffer(sc)
    for n in range(4):                                   # preview.cuh:306 calling pattern
        crt.render(sc, a, samples=1, samples_done=n)
        crt.render_static(sc, b, samples=1, samples_done=n)
    _assert_same_bits(a.download(), b.download())
    a.close(); b.close(); sc.close()


def test_static_scratch_full_size_frame_within_summation_order(crt):
    # 1000x1000: crt_render cuts 8 samples into 4 chunks, the 64 MiB array holds one chunk, so the
    # per-pixel sums are grouped differently: same samples, float summation order differs.
    (ref, ref_rays), (got, got_rays) = _render_both(crt, "cornell_box.yaml", 1000, 1000, 8)
    assert got_rays == ref_rays
    for k in oracle_lib.PASSES:
        assert np.array_equal(np.isnan(ref[k]), np.isnan(got[k])), k
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-5, atol=1e-6, equal_nan=True, err_msg=k)


def test_static_scratch_rejects_images_beyond_its_size(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
    sc.set_render(1280, 1024, samples=1)
    buf = crt.create_buffer(sc)
    with pytest.raises(crt.CrtError) as e:
        crt.render_static(sc, buf)
    assert e.value.status == -5 and "exceed the scratch available" in str(e.value)
    crt.render(sc, buf)                                  # the ordinary entry point is unaffected
    buf.close(); sc.close()


def _free_device_bytes():
    """hipMemGetInfo of the current device, straight from the HIP runtime the library itself uses."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_static_scratch_call_allocates_no_device_memory(crt):
    """What the entry point is for: between the start and the end of a call the device's free memory does not move
    (a hipMalloc inside the call, even one that is freed again before it returns, would show in the middle reading)."""
    import threading
    sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
    sc.set_render(640, 480, samples=16)
    buf = crt.create_buffer(sc)
    crt.render_static(sc, buf)                           # warmed up: code object loaded, scratch address known
    before = _free_device_bytes()
    lowest = [before]
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            lowest[0] = min(lowest[0], _free_device_bytes())

    t = threading.Thread(target=watch)
    t.start()
    try:
        for _ in range(5):
            crt.render_static(sc, buf)
    finally:
        stop.set()
        t.join()
    assert _free_device_bytes() == before and lowest[0] == before, (before, lowest[0], _free_device_bytes())
    buf.close(); sc.close()


def test_overlapping_static_calls_on_one_gpu_are_refused_not_mixed(crt):
    """Calls on one GPU share one scratch array: a call that arrives while another is running gets an error; every
    call that does return an image returns the right one."""
    import threading
    names = ["cornell_box.yaml", "orthographic.yaml", "cornell_box.yaml", "orthographic.yaml"]
    scenes, bufs, want = [], [], []
    for name in names:
        sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
        sc.set_render(160, 120, samples=6)
        b = crt.create_buffer(sc)
        crt.render(sc, b)
        want.append(b.download())
        scenes.append(sc); bufs.append(b)
    outcomes = [[] for _ in names]
    barrier = threading.Barrier(len(names))

    def work(k):
        for _ in range(6):
            barrier.wait()
            try:
                crt.render_static(scenes[k], bufs[k])
                got = bufs[k].download()
                outcomes[k].append(all(np.array_equal(got[p].view(np.uint32), want[k][p].view(np.uint32)) for p in oracle_lib.PASSES))
            except crt.CrtError as e:
                outcomes[k].append("another crt_render_static call is running" in str(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(names))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(all(o) and len(o) == 6 for o in outcomes), outcomes
    for b in bufs:
        b.close()
    for sc in scenes:
        sc.close()
