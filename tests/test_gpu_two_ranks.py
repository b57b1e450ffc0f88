"""Two processes on ONE GPU, each rendering its share of a frame with the HIP library exactly as a
rank of bench.py does (crt_render_range for the sample split, crt_render_bands for the tile split),
joined by the same multi_gpu helpers over gloo (RCCL does not accept two ranks on one device, so
the device buffers are copied to the host for the collective).  The combined image must equal the
single-process device render: within float summation order for the sample split (each rank's mean
is scaled and summed), bit for bit for the tile split."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib
from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

SIZE_W, SIZE_H = 104, 100


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, scene, spp, how, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import multi_gpu
    torch.cuda.set_device(0)
    crt.set_device(0)
    sc = crt.load_scene(scene_path(scene), base_dir=scene_path(""))
    sc.set_render(SIZE_W, SIZE_H, samples=spp)
    sc.upload()
    if how == "samples":
        first, count = multi_gpu.split_samples(spp, world)[rank]
        dev = torch.zeros(14 * SIZE_W * SIZE_H, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        buf = crt.Buffer(SIZE_W, SIZE_H, device_ptr=dev.data_ptr())
        crt.render_launch(sc, buf, first, count)
        _, rays = crt.render_finish(sc)
        host = dev.cpu()
        multi_gpu.combine_means_async(host, count, spp, dist).wait()
        image = host.numpy()
    else:
        rows = crt.band_row_count(SIZE_H, multi_gpu.BAND_ROWS, rank, world)
        stride = multi_gpu.band_stride_floats(SIZE_W, SIZE_H, world)
        dev = torch.full((stride,), -7.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        own = crt.Buffer(SIZE_W, rows, device_ptr=dev.data_ptr())
        crt.render_bands_launch(sc, own, 0, spp, multi_gpu.BAND_ROWS, rank, world)
        _, rays = crt.render_finish(sc)
        gathered = torch.empty(world * stride)
        multi_gpu.gather_bands_async(dev.cpu(), gathered, dist).wait()
        # the device assemble kernel on the gathered block, as bench.py runs it after the all-gather
        g_dev = gathered.cuda()
        full_dev = torch.zeros(14 * SIZE_W * SIZE_H, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        full = crt.Buffer(SIZE_W, SIZE_H, device_ptr=full_dev.data_ptr())
        crt.assemble_bands(full, g_dev.data_ptr(), stride, multi_gpu.BAND_ROWS, world,
                           stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        image = full_dev.cpu().numpy()
        full.close()
        own.close()
    total = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(total)
    np.save(os.path.join(out_dir, f"image{rank}.npy"), image)
    if rank == 0:
        np.save(os.path.join(out_dir, "rays.npy"), total.numpy())
    sc.close()
    dist.destroy_process_group()


def _single(crt, scene, spp):
    from cudaraytracer_amd import api
    sc = crt.load_scene(scene_path(scene), base_dir=scene_path(""))
    sc.set_render(SIZE_W, SIZE_H, samples=spp)
    whole = crt.create_buffer(sc)
    crt.render(sc, whole)
    ref, rays = whole.download(), api.last_ray_count
    whole.close()
    sc.close()
    return np.concatenate([ref[p].ravel() for p in oracle_lib.PASSES]), rays


@pytest.mark.parametrize("scene,spp", [("cornell_box.yaml", 5), ("obj.yaml", 3)])
def test_two_processes_sample_split_on_the_device(crt, tmp_path, scene, spp):
    world = 2
    ref, ref_rays = _single(crt, scene, spp)
    mp.spawn(_rank, args=(world, _free_port(), scene, spp, "samples", str(tmp_path)), nprocs=world, join=True)
    assert int(np.load(tmp_path / "rays.npy")[0]) == ref_rays
    for rank in range(world):
        got = np.load(tmp_path / f"image{rank}.npy")
        assert np.array_equal(np.isnan(got), np.isnan(ref))            # missed depth samples stay NaN
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6, equal_nan=True)


@pytest.mark.parametrize("scene,spp", [("cornell_box.yaml", 3), ("obj.yaml", 2)])
def test_two_processes_tile_split_on_the_device(crt, tmp_path, scene, spp):
    world = 2
    ref, ref_rays = _single(crt, scene, spp)
    mp.spawn(_rank, args=(world, _free_port(), scene, spp, "tiles", str(tmp_path)), nprocs=world, join=True)
    assert int(np.load(tmp_path / "rays.npy")[0]) == ref_rays
    for rank in range(world):
        got = np.load(tmp_path / f"image{rank}.npy")
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
