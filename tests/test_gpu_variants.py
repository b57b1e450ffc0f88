"""The kernel variant a scene runs is an implementation choice: the variant compiled for just the
scene's features against the one compiled for every feature, and box tests of cuboid slots taken
together against every slot on its own, must give the same image bit for bit and the same ray
count (single samples, so that no sum depends on the order in which paths finish)."""
import os

import numpy as np
import pytest

from conftest import scene_path

pytestmark = pytest.mark.gpu

SCENE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")

SCENES = [("cornell_box.yaml", 160, 160), ("orthographic.yaml", 128, 128), ("materials.yaml", 200, 40),
          ("features_small.yaml", 96, 96), ("features_tree.yaml", 96, 96), ("obj.yaml", 96, 96)]


def _render(crt, name, w, h, options):
    from cudaraytracer_amd import api
    sc = crt.load_scene(scene_path(name), base_dir=SCENE_DIR)
    sc.set_render(w, h, samples=4)
    sc.set_options(**options)
    buf = crt.create_buffer(sc)
    frames, rays = [], []
    for done in range(3):
        crt.render(sc, buf, samples=1, samples_done=done)
        rays.append(api.last_ray_count)
        frames.append(buf.download())
    text = sc.describe()
    buf.close()
    sc.close()
    return frames, rays, text


@pytest.mark.parametrize("name,w,h", SCENES)
def test_variants_agree(crt, name, w, h):
    ref, ref_rays, ref_text = _render(crt, name, w, h, {})
    for options in ({"general_kernel": 1}, {"flat_cuboids": 2}, {"general_kernel": 1, "flat_cuboids": 2}):
        got, rays, text = _render(crt, name, w, h, options)
        assert rays == ref_rays, f"{name} {options}: ray counts {rays} != {ref_rays} ({text} vs {ref_text})"
        for a, b in zip(got, ref):
            for k in a:
                same = (a[k].view(np.uint32) == b[k].view(np.uint32)) | (np.isnan(a[k]) & np.isnan(b[k]))
                assert same.all(), f"{name} {options}: pass {k} differs at {np.flatnonzero(~same)[:8]} ({text} vs {ref_text})"


def test_cornell_runs_the_specialised_variant(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    sc.upload()
    text = sc.describe()
    sc.close()
    assert "mode=flat" in text and "features=0x100" in text and "cuboids=3" in text, text
    sc = crt.load_scene(scene_path("cornell_box.yaml"))
    sc.set_options(general_kernel=1, flat_cuboids=2)
    sc.upload()
    text = sc.describe()
    sc.close()
    assert "features=0x1ff" in text and "cuboids=0" in text, text
