"""The wavefront pipeline (generate / trace / shade per bounce / accumulate, path state in HBM) against
the fused kernel and the oracle: same device functions, so every pass of every sample and the ray
count agree bit for bit; bulk calls add samples in sample order and are reproducible."""
import numpy as np
import pytest

import oracle_lib
from conftest import scene_path

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    return all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in oracle_lib.PASSES)


def _values_equal(a, b):
    """Equal as numbers, NaN in the same places: the device's sums start at +0, so a -0 component of
    the oracle comes back as +0."""
    return all(((a[k] == b[k]) | (np.isnan(a[k]) & np.isnan(b[k]))).all() for k in oracle_lib.PASSES)


@pytest.mark.parametrize("name", ["obj.yaml", "features_tree.yaml", "cornell_box.yaml", "nested_wrappers.yaml"])
def test_wavefront_bit_exact_per_sample(crt, oracle, name):
    from cudaraytracer_amd import api
    w, h = 104, 72
    sc = crt.load_scene(scene_path(name), base_dir=scene_path(""))
    sc.set_render(w, h, samples=3)
    sc.set_options(wavefront=1)
    sc.set_mode(1)
    assert "pipeline=wavefront" in sc.describe()
    dp = crt.lib().crt_host_scene_desc(sc._host)
    buf = crt.create_buffer(sc)
    ref = oracle_lib.new_passes(w, h)
    oracle.oracle_set_pow5_mulchain(1)
    try:
        for done in range(3):                                  # the preview's calling pattern, sample by sample
            crt.render(sc, buf, samples=1, samples_done=done)
            rays = api.last_ray_count
            _, stats = oracle_lib.render(oracle, dp, w, h, samples=1, samples_done=done, passes=ref)
            assert rays == stats["rays"], done
            assert _values_equal(buf.download(), ref), done
    finally:
        oracle.oracle_set_pow5_mulchain(0)
    buf.close()
    sc.close()


def test_wavefront_bulk_call_batches_and_reproducibility(crt):
    """Several samples per call, in batches smaller than the call (wavefront_paths holds two samples of
    the image): equal to the fused kernel's render to float summation order, the ray counts equal,
    and two runs equal in every bit."""
    from cudaraytracer_amd import api
    w, h, spp = 200, 120, 7
    images, rays = [], []
    for options in ({"wavefront": 2}, {"wavefront": 1, "wavefront_paths": 2 * 200 * 120}, {"wavefront": 1, "wavefront_paths": 2 * 200 * 120}):
        sc = crt.load_scene(scene_path("obj.yaml"), base_dir=scene_path(""))
        sc.set_render(w, h, samples=spp)
        sc.set_options(**options)
        sc.set_mode(1)
        assert ("pipeline=wavefront" in sc.describe()) == (options["wavefront"] == 1)
        buf = crt.create_buffer(sc)
        crt.render(sc, buf)
        images.append(buf.download())
        rays.append(api.last_ray_count)
        buf.close()
        sc.close()
    assert rays[0] == rays[1] == rays[2]
    assert _bits_equal(images[1], images[2])
    for k in oracle_lib.PASSES:
        assert np.array_equal(np.isnan(images[0][k]), np.isnan(images[1][k])), k
        np.testing.assert_allclose(images[1][k], images[0][k], rtol=2e-5, atol=1e-6, equal_nan=True, err_msg=k)


def test_wavefront_is_off_unless_asked_for(crt):
    sc = crt.load_scene(scene_path("obj.yaml"), base_dir=scene_path(""))
    assert "pipeline=wavefront" not in sc.describe()
    sc.close()
