"""The order in which a frame's work items (pixel tile x sample chunk) are handed to the waves is a scheduling
choice: natural order, most expensive first by the measured cost of the last launch, or the library's measured
choice between the two must give the same image bit for bit and the same ray count, launch after launch (single
samples per launch, so that no sum depends on the order in which paths finish).  An order that dropped or doubled
an item would change the ray count."""
import os

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

SCENE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
LAUNCHES = 9        # the measured choice alternates for six launches, then settles


def _render(crt, name, w, h, options, samples_per_launch=1):
    from cudaraytracer_amd import api
    sc = crt.load_scene(scene_path(name), base_dir=SCENE_DIR)
    sc.set_render(w, h, samples=LAUNCHES * samples_per_launch)
    sc.set_options(**options)
    buf = crt.create_buffer(sc)
    frames, rays = [], []
    for k in range(LAUNCHES):
        crt.render(sc, buf, samples=samples_per_launch, samples_done=k * samples_per_launch)
        rays.append(api.last_ray_count)
        frames.append(buf.download())
    buf.close()
    sc.close()
    return frames, rays


@pytest.mark.parametrize("name,w,h", [("cornell_box.yaml", 160, 160), ("materials.yaml", 400, 80), ("obj.yaml", 96, 96), ("features_small.yaml", 96, 96)])
def test_item_order_does_not_change_the_image(crt, name, w, h):
    ref, ref_rays = _render(crt, name, w, h, {"item_order": 1})
    for options in ({"item_order": 2}, {}):
        got, rays = _render(crt, name, w, h, options)
        assert rays == ref_rays, f"{name} {options}: ray counts {rays} != {ref_rays}"
        for launch, (a, b) in enumerate(zip(got, ref)):
            for k in a:
                same = (a[k].view(np.uint32) == b[k].view(np.uint32)) | (np.isnan(a[k]) & np.isnan(b[k]))
                assert same.all(), f"{name} {options} launch {launch}: pass {k} differs at {np.flatnonzero(~same)[:8]}"


def test_item_order_with_many_samples_per_launch(crt):
    """Chunks of several samples (sums depend on the order paths finish in, so only close, not equal): every launch of
    either order traces the same rays and gives the same image within the run-to-run spread."""
    ref, ref_rays = _render(crt, "materials.yaml", 400, 80, {"item_order": 1}, samples_per_launch=12)
    got, rays = _render(crt, "materials.yaml", 400, 80, {"item_order": 2}, samples_per_launch=12)
    assert rays == ref_rays
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a["color"], b["color"], rtol=2e-4, atol=2e-5)


def test_item_order_option_range(crt):
    from cudaraytracer_amd.api import CrtError
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    sc.set_options(item_order=3)
    with pytest.raises(CrtError, match="item_order"):
        sc.upload()
    sc.close()


def test_renders_side_by_side_on_parts_of_the_chip(crt):
    """chip_parts = 3: a render's persistent kernel takes a third of the GPU, for callers that keep three renders in
    flight on three copies of the scene (the N-rank bench).  The image of each is the one a render alone on the whole
    chip gives, bit for bit (one sample per launch), and so is the ray count."""
    from cudaraytracer_amd import api
    name, w, h = "cornell_box.yaml", 160, 160
    ref, ref_rays = _render(crt, name, w, h, {"item_order": 1})
    scenes, bufs = [], []
    for _ in range(3):
        sc = crt.load_scene(scene_path(name), base_dir=SCENE_DIR)
        sc.set_render(w, h, samples=LAUNCHES)
        sc.set_options(chip_parts=3, item_order=1)
        sc.upload()
        scenes.append(sc)
        bufs.append(crt.create_buffer(sc))
    for sc, buf in zip(scenes, bufs):                       # sample number 1 of every pixel, three times over, together
        api.render_launch(sc, buf, 0, 1)
    for sc, buf in zip(scenes, bufs):
        _, rays = api.render_finish(sc)
        assert rays == ref_rays[0]
        got = buf.download()
        for k in got:
            same = (got[k].view(np.uint32) == ref[0][k].view(np.uint32)) | (np.isnan(got[k]) & np.isnan(ref[0][k]))
            assert same.all(), f"pass {k} differs at {np.flatnonzero(~same)[:8]}"
    for sc, buf in zip(scenes, bufs):
        buf.close()
        sc.close()
