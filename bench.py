#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on the Cornell box
(1000x1000, 50 spp, 15 bounces) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one full render of the scene per GPU.  With N GPUs rank k renders sample numbers
k*spp+1 .. (k+1)*spp of every pixel (weak scaling: per-GPU work is fixed, the image gets N*spp
samples) and the ranks' means are combined by one RCCL all-reduce of the 14*W*H pass floats,
which runs while the next frame renders (every all-reduce completes inside the timed region).
Inputs (scene in HBM, pass buffers allocated) are resident before the timed region; the timed
region contains the render kernels, the resolve kernel and the all-reduce.

A ray = one closest-hit query = one call of bvh4_hit at the reference's render.cuh:47; the
kernels count them on the device.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic work per ray under the reference traversal order (survey constants):
#   flops/ray F = 24*N_box + 84*N_prim + 60*N_accept + 100, bytes/ray B = 128 (path state + hit record)
FLOPS_PER_RAY = {"cornell_box.yaml": 1116.0, "orthographic.yaml": 485.0, "obj.yaml": 1488.0}
BYTES_PER_RAY = 128.0
PEAK_FP32_TFLOPS = 157.3     # MI355X vector FP32 peak
PEAK_HBM_GBS = 8000.0        # MI355X HBM3E peak


def measured_traffic(scene):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (separate
    rocprofv3 --pmc runs, corrected as the gfx950 notes prescribe); None if not measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f)[scene]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(scene_file, width, height, spp, threads):
    """Oracle (CPU restatement of the reference) timed on the host cores on a bounded sample of
    the same workload: the full-size image at a reduced sample count."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import cudaraytracer_amd as crt
    lib = oracle_lib.load()
    sc = crt.load_scene(scene_file)
    sc.set_render(width, height, samples=spp)
    dp = crt.lib().crt_host_scene_desc(sc._host)
    t0 = time.perf_counter()
    _, stats = oracle_lib.render(lib, dp, width, height, samples=spp, threads=threads)
    dt = time.perf_counter() - t0
    sc.close()
    return {"value": round(stats["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{os.path.basename(scene_file)} {width}x{height}, {spp} of the spp, all bounces: "
                      f"{stats['rays']} rays in {dt:.2f} s on {threads} OpenMP threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="cornell_box.yaml")
    ap.add_argument("--cpu-spp", type=int, default=10, help="samples per pixel of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    # stdout carries exactly one line, the result; anything native libraries print on the way
    # (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import multi_gpu

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    crt.set_device(local_rank)
    # under torch.distributed.run (RANK is set) the collective path runs even with one rank, so a
    # single-GPU box can rehearse exactly what the N-GPU launch does
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    scene_file = os.path.join(ROOT, "scenes", args.scene)
    sc = crt.load_scene(scene_file)
    ri = sc.render_info
    W, H, spp, bounces = ri.image_width, ri.image_height, ri.samples_per_pixel, ri.max_bounces
    sc.upload()
    # Frames in flight: with more than one process the all-reduce of frame k runs while frame k+1
    # (and the start of k+2) renders, so three sets of pass buffers rotate; a frame's buffers are
    # rendered into again only after its all-reduce has finished.
    depth = 3 if distributed else 1
    passes = [torch.zeros(14 * W * H, dtype=torch.float32, device="cuda") for _ in range(depth)]
    bufs = [crt.Buffer(W, H, ri.gamma, device_ptr=p.data_ptr()) for p in passes]
    lib = crt.lib()
    dev = sc.get_device_scene()
    first, count = multi_gpu.rank_sample_range(rank, world, spp)
    pending = [None] * depth

    def wait_for(slot):
        if pending[slot] is not None:
            pending[slot].wait()
            torch.cuda.current_stream().synchronize()
            pending[slot] = None

    def run(steps):
        """`steps` frames; returns (rays of this rank, summed kernel ms, summed resolve ms)."""
        rays = 0
        trace_ms = resolve_ms = 0.0
        previous = None
        for k in range(steps):
            slot = k % depth
            wait_for(slot)
            crt.render_launch(sc, bufs[slot], first, count)
            if previous is not None and distributed:
                pending[previous] = multi_gpu.combine_means_async(passes[previous], count, spp * world, dist)
            _, r = crt.render_finish(sc)
            tr, rs = C.c_float(0), C.c_float(0)
            lib.crt_render_timings(dev, C.byref(tr), C.byref(rs))
            rays += r
            trace_ms += tr.value
            resolve_ms += rs.value
            previous = slot
        if previous is not None and distributed:
            pending[previous] = multi_gpu.combine_means_async(passes[previous], count, spp * world, dist)
        for slot in range(depth):
            wait_for(slot)
        return rays, trace_ms, resolve_ms

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    rays_rank, trace_ms, resolve_ms = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0

    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rr = torch.tensor([rays_rank], dtype=torch.int64, device="cuda")
        dist.all_reduce(rr)
        rays_total = int(rr.item())
    else:
        rays_total = rays_rank

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        rays_per_launch = rays_rank / args.steps
        kernel_ms = trace_ms / args.steps
        flops = FLOPS_PER_RAY.get(args.scene)
        roofline = None
        if flops:
            achieved = rays_per_launch * flops / (kernel_ms * 1e-3) / 1e12
            hbm = rays_per_launch * BYTES_PER_RAY / (kernel_ms * 1e-3) / 1e9
            roofline = {"bound": "valu", "kernel": "render_fused", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": measured_traffic(args.scene),
                        "kernel_ms": round(kernel_ms, 4), "rays_per_launch": int(rays_per_launch),
                        "flops_per_ray": flops,
                        "hbm": {"achieved": round(hbm, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(hbm / PEAK_HBM_GBS, 4), "bytes_per_ray": BYTES_PER_RAY}}
        out = {
            "metric": "Mrays/s (whole node) Cornell box 1000x1000, 50spp, 15 bounces" if args.scene == "cornell_box.yaml"
                      else f"Mrays/s (whole node) {args.scene}",
            "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene} {W}x{H}, {spp} spp per GPU, {bounces} bounces",
                       "sharding": "sample ranges per rank, one all-reduce of 14*W*H floats per frame, overlapped with the next frame" if world > 1 else "single GPU",
                       "rays_per_step": rays_total // args.steps},
            "roofline": roofline,
        }
        if world == 1 and args.cpu_spp > 0:
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(scene_file, W, H, args.cpu_spp, threads)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    for b in bufs:
        b.close()
    sc.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
