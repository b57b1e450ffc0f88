"""Sample-range sharding (the N>1 path of bench.py) rehearsed with 2 ranks on CPU/gloo: each
rank renders its sample range with the oracle, ranks combine with combine_means(); the result
must equal a one-process render of all samples up to float summation order."""
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


def _worker(rank, world, port, size, spp, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import multi_gpu
    lib = oracle_lib.load()
    sc = crt.load_scene(scene_path("orthographic.yaml"))
    sc.set_render(size, size, samples=spp * world)
    first, count = multi_gpu.rank_sample_range(rank, world, spp)
    mean = _mean_of_range(crt, lib, sc, size, first, count)
    t = torch.from_numpy(mean.astype(np.float32))
    t_async = t.clone()
    multi_gpu.combine_means(t, count, spp * world, dist)
    # the overlapped form bench.py uses: start, do other work, wait
    work = multi_gpu.combine_means_async(t_async, count, spp * world, dist)
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


def test_split_helpers():
    from cudaraytracer_amd import multi_gpu
    assert multi_gpu.rank_sample_range(3, 8, 50) == (150, 50)
    parts = multi_gpu.split_samples(1024, 8)
    assert parts[0] == (0, 128) and parts[-1] == (896, 128)
    parts = multi_gpu.split_samples(10, 4)
    assert [n for _, n in parts] == [3, 3, 2, 2] and parts[3][0] == 8
    with pytest.raises(ValueError):
        multi_gpu.rank_sample_range(2, 2, 1)
