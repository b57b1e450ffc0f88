"""The N>1 paths of bench.py rehearsed with 2 ranks on CPU/gloo.
Sample split (weak: equal ranges; strong: the scene's samples divided, unevenly if need be): each
rank renders its sample range with the oracle, ranks combine with combine_means(); the result must
equal a one-process render of all samples up to float summation order.
Tile split: each rank holds the rows of its bands, gather_bands_async() + the reference form of the
assemble kernel must give back the whole image exactly."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _mean_of_range(crt, lib, sc, size, first, count):
    """Mean over sample numbers first+1 .. first+count with the oracle (samples folded one call
    at a time from an empty buffer, like crt_render_range does on the device)."""
    dp = crt.lib().crt_host_scene_desc(sc._host)
    acc = None
    for k in range(count):
        one = oracle_lib.new_passes(size, size)
        # a fresh buffer with samples_done = first+k would fold into zeros; render sample number
        # first+k+1 alone by asking for it as the "first" sample of a scene whose RNG key is the
        # global sample number: oracle_render keys the RNG by samples_done + s + 1
        oracle_lib.render(lib, dp, size, size, samples=1, samples_done=first + k, passes=one, threads=2)
        # undo the fold of the single sample into the zero buffer: value = mean * n
        n = float(first + k + 1)
        flat = np.concatenate([one[p].ravel() for p in oracle_lib.PASSES]) * (n if first + k > 0 else 1.0)
        acc = flat if acc is None else acc + flat
    return acc / count


def _worker(rank, world, port, size, spp, out_dir, strong=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import multi_gpu
    lib = oracle_lib.load()
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    total = spp if strong else spp * world
    sc.set_render(size, size, samples=total)
    first, count = multi_gpu.split_samples(total, world)[rank] if strong else multi_gpu.rank_sample_range(rank, world, spp)
    mean = _mean_of_range(crt, lib, sc, size, first, count)
    t = torch.from_numpy(mean.astype(np.float32))
    t_async = t.clone()
    multi_gpu.combine_means(t, count, total, dist)
    # the overlapped form bench.py uses: start, do other work, wait
    work = multi_gpu.combine_means_async(t_async, count, total, dist)
    assert work is not None
    work.wait()
    assert torch.equal(torch.nan_to_num(t), torch.nan_to_num(t_async))
    if rank == 0:
        np.save(os.path.join(out_dir, "combined.npy"), t.numpy())
    dist.destroy_process_group()


def test_two_rank_sample_sharding_matches_single_process(crt, oracle, tmp_path):
    size, spp, world = 24, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), size, spp, str(tmp_path)), nprocs=world, join=True)
    combined = np.load(tmp_path / "combined.npy")
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    sc.set_render(size, size, samples=spp * world)
    ref, _ = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), size, size)
    flat = np.concatenate([ref[p].ravel() for p in oracle_lib.PASSES])
    assert np.array_equal(np.isnan(combined), np.isnan(flat))        # NaN depth survives the reduction
    np.testing.assert_allclose(combined, flat, rtol=2e-5, atol=1e-6, equal_nan=True)


def test_two_rank_strong_split_of_an_odd_sample_count(crt, oracle, tmp_path):
    """bench.py's default for N > 1: the scene's own sample count divided over the ranks (3 = 2 + 1),
    means weighted by each rank's share."""
    size, spp, world = 24, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), size, spp, str(tmp_path), True), nprocs=world, join=True)
    combined = np.load(tmp_path / "combined.npy")
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    sc.set_render(size, size, samples=spp)
    ref, _ = oracle_lib.render(oracle, crt.lib().crt_host_scene_desc(sc._host), size, size)
    flat = np.concatenate([ref[p].ravel() for p in oracle_lib.PASSES])
    assert np.array_equal(np.isnan(combined), np.isnan(flat))
    np.testing.assert_allclose(combined, flat, rtol=2e-5, atol=1e-6, equal_nan=True)


def _band_worker(rank, world, port, width, height, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cudaraytracer_amd import multi_gpu
    full = _synthetic_passes(width, height)
    rows = multi_gpu.band_rows_of_rank(height, rank, world)
    stride = multi_gpu.band_stride_floats(width, height, world)
    own = torch.full((stride,), -7.0)                                  # padding must not leak into the image
    n_full, parts = width * height, []
    for k in range(4):
        parts.append(full[3 * k * n_full:3 * (k + 1) * n_full].reshape(height, width, 3)[rows].ravel())
    for k in range(2):
        parts.append(full[(12 + k) * n_full:(13 + k) * n_full].reshape(height, width)[rows].ravel())
    packed = np.concatenate(parts)
    own[:packed.size] = torch.from_numpy(packed)
    gathered = torch.empty(world * stride)
    multi_gpu.gather_bands_async(own, gathered, dist).wait()
    image = multi_gpu.assemble_bands_reference(gathered.numpy(), width, height, world)
    np.save(os.path.join(out_dir, f"image{rank}.npy"), image)
    dist.destroy_process_group()


def _synthetic_passes(width, height):
    g = np.random.default_rng(11)
    full = g.standard_normal(14 * width * height).astype(np.float32)
    full[12 * width * height + 5] = np.nan                            # a missed depth sample
    return full


@pytest.mark.parametrize("height", [40, 37])                          # whole bands / a ragged last band
def test_two_rank_tile_split_gathers_the_whole_image(tmp_path, height):
    width, world = 12, 2
    mp.spawn(_band_worker, args=(world, _free_port(), width, height, str(tmp_path)), nprocs=world, join=True)
    full = _synthetic_passes(width, height)
    for rank in range(world):
        image = np.load(tmp_path / f"image{rank}.npy")
        assert np.array_equal(image.view(np.uint32), full.view(np.uint32))


def test_sharding_choice():
    from cudaraytracer_amd import multi_gpu
    # 50 samples do not divide over 8 GPUs (7 against 6.25): the tile split wins
    assert multi_gpu.choose_sharding(1000, 1000, 50, 8) == "tiles"
    assert multi_gpu.choose_sharding(1000, 1000, 50, 1) == "samples"
    assert multi_gpu.choose_sharding(64, 32, 64, 8) == "samples"       # too few rows for 8 bands
    assert multi_gpu.choose_sharding(1000, 1000, 4, 8) == "tiles"      # fewer samples than GPUs
    # rows of the ranks partition the image
    rows = sorted(j for r in range(3) for j in multi_gpu.band_rows_of_rank(37, r, 3))
    assert rows == list(range(37))
    # a slow frame makes the samples split's imbalance dearer, not cheaper: still tiles
    assert multi_gpu.choose_sharding(3840, 2160, 50, 8, single_gpu_ms=360.0) == "tiles"
    # 1024 samples divide evenly over 8 GPUs, 9 bands do not (2 against 1.125): samples, once the frame
    # is slow enough to outweigh the all-reduce's extra bytes
    assert multi_gpu.choose_sharding(1000, 72, 1024, 8, single_gpu_ms=100.0) == "samples"
    assert multi_gpu.choose_sharding(1000, 72, 1024, 8, single_gpu_ms=0.01) == "tiles"
    t8 = multi_gpu.predicted_frame_ms("tiles", 11.5, 1000, 1000, 50, 8)
    s8 = multi_gpu.predicted_frame_ms("samples", 11.5, 1000, 1000, 50, 8)
    assert t8 < s8 < 11.5


def test_sharding_that_leaves_a_rank_without_work_is_refused_alike_on_every_rank():
    """Feasibility is a pure function of (frame, world): every rank raises the same error before any
    collective, none is left waiting for a rank that stopped on its own."""
    from cudaraytracer_amd import multi_gpu
    # 4 samples over 8 GPUs: ranks 4..7 would get none
    assert not multi_gpu.sharding_is_feasible("samples", 1000, 4, 8)
    with pytest.raises(ValueError, match="samples"):
        multi_gpu.check_sharding("samples", 1000, 4, 8)
    # 32 rows = 4 bands over 8 GPUs: ranks 4..7 would own no rows
    assert not multi_gpu.sharding_is_feasible("tiles", 32, 64, 8)
    with pytest.raises(ValueError, match="bands"):
        multi_gpu.check_sharding("tiles", 32, 64, 8)
    # 57 rows = 8 bands (the last one ragged): every rank owns rows
    assert multi_gpu.sharding_is_feasible("tiles", 57, 1, 8)
    assert all(multi_gpu.band_rows_of_rank(57, r, 8) for r in range(8))
    multi_gpu.check_sharding("tiles", 57, 1, 8)
    assert multi_gpu.choose_sharding(64, 57, 4, 8) == "tiles"
    # neither split works: automatic choice refuses too (it used to answer "samples")
    with pytest.raises(ValueError, match="neither"):
        multi_gpu.choose_sharding(64, 32, 4, 8)
    # the feasibility answer agrees with what the splits hand out
    for spp, height, world in ((4, 32, 8), (50, 1000, 8), (8, 64, 8), (3, 24, 2)):
        assert multi_gpu.sharding_is_feasible("samples", height, spp, world) == all(n > 0 for _, n in multi_gpu.split_samples(spp, world))
        assert multi_gpu.sharding_is_feasible("tiles", height, spp, world) == all(
            multi_gpu.band_rows_of_rank(height, r, world) for r in range(world))


def test_split_helpers():
    from cudaraytracer_amd import multi_gpu
    assert multi_gpu.rank_sample_range(3, 8, 50) == (150, 50)
    parts = multi_gpu.split_samples(1024, 8)
    assert parts[0] == (0, 128) and parts[-1] == (896, 128)
    parts = multi_gpu.split_samples(10, 4)
    assert [n for _, n in parts] == [3, 3, 2, 2] and parts[3][0] == 8
    with pytest.raises(ValueError):
        multi_gpu.rank_sample_range(2, 2, 1)
