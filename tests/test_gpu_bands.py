"""Tile split over GPUs (crt_render_bands + crt_buffers_assemble_bands) rehearsed on one GPU: the
bands of all ranks are rendered one after the other, laid out like an all-gather would leave them
and assembled on the device; the result must be the single-GPU image bit for bit (a pixel gets the
same samples whichever rank renders it) and the ray counts must add up."""
import numpy as np
import pytest

import oracle_lib
from conftest import scene_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,mode,scene", [(2, 0, "cornell_box.yaml"), (3, 0, "orthographic.yaml"), (8, 0, "cornell_box.yaml"),
                                              (3, 1, "obj.yaml"), (8, 2, "features_small.yaml")])
def test_bands_assemble_to_the_single_gpu_image(crt, world, mode, scene):
    import torch
    from cudaraytracer_amd import api, multi_gpu
    width, height, spp = 104, 100, 3                      # 100 rows: the last band is ragged, ranks own different row counts
    sc = crt.load_scene(scene_path(scene), base_dir=scene_path(""))
    sc.set_render(width, height, samples=spp)
    sc.set_mode(mode)
    whole = crt.create_buffer(sc)
    crt.render(sc, whole)
    ref, ref_rays = whole.download(), api.last_ray_count
    whole.close()

    stride = multi_gpu.band_stride_floats(width, height, world)
    gathered = torch.full((world * stride,), -7.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rays = 0
    for rank in range(world):
        rows = crt.band_row_count(height, multi_gpu.BAND_ROWS, rank, world)
        assert rows == len(multi_gpu.band_rows_of_rank(height, rank, world))
        own = crt.Buffer(width, rows, device_ptr=gathered[rank * stride:].data_ptr())
        _, r = crt.render_bands(sc, own, 0, spp, multi_gpu.BAND_ROWS, rank, world)
        rays += r
        own.close()
    assert rays == ref_rays
    image = torch.zeros(14 * width * height, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    full = crt.Buffer(width, height, device_ptr=image.data_ptr())
    crt.assemble_bands(full, gathered.data_ptr(), stride, multi_gpu.BAND_ROWS, world, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = full.download()
    for k in oracle_lib.PASSES:
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
    # the CPU form of the assemble step (used by the gloo test) agrees with the kernel
    flat = multi_gpu.assemble_bands_reference(gathered.cpu().numpy(), width, height, world)
    assert np.array_equal(flat.view(np.uint32), image.cpu().numpy().view(np.uint32))
    full.close()
    sc.close()


def test_band_arguments_are_checked(crt):
    sc = crt.load_scene(scene_path("cornell_box.yaml"), base_dir=scene_path(""))
    sc.set_render(64, 64, samples=1)
    whole = crt.create_buffer(sc)
    with pytest.raises(crt.CrtError, match="band's buffers hold its own 32 rows"):
        crt.render_bands(sc, whole, 0, 1, 8, 0, 2)       # a band buffer has the band's rows, not the image's
    own = crt.Buffer(64, 32)
    with pytest.raises(crt.CrtError, match="multiple of 8"):
        crt.render_bands(sc, own, 0, 1, 12, 0, 2)
    with pytest.raises(crt.CrtError, match="band_index"):
        crt.render_bands(sc, own, 0, 1, 8, 2, 2)
    crt.render_bands(sc, own, 0, 1, 8, 1, 2)
    own.close(); whole.close(); sc.close()
