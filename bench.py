#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the path-tracing hot path on the Cornell box
(1000x1000, 50 spp, 15 bounces) on N MI355X of one node.

  python bench.py --gpus N --steps K --warmup W [--scene S] [--scaling strong|weak] [--sharding auto|samples|tiles]

N > 1 runs one process per GPU.  Started under a launcher (python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py --gpus N ..., RANK / LOCAL_RANK / WORLD_SIZE in the
environment) this process is one rank.  Started plainly (no RANK in the environment) with
--gpus N > 1 it is the PARENT: before anything touches a GPU it checks that N devices are visible,
starts the N ranks as child processes under torch.distributed.run, relays rank 0's JSON line and
exits with the children's status.  A launcher's WORLD_SIZE that differs from --gpus is an error.

A step = one full render of the scene's frame.  With N GPUs the default is STRONG scaling: the
same frame (the scene's own sample count) is divided over the ranks, either by image tiles (row
bands dealt round-robin, every rank renders all samples of its rows, one all-gather + a copy
kernel) or by sample ranges (every rank renders part of the samples of every pixel, one all-reduce
of the 14*W*H pass floats); `--sharding auto` picks by predicted cost (multi_gpu.choose_sharding).
`--scaling weak` gives every rank the scene's full sample count (the image gets N times the
samples).  The collective of frame k runs while frame k+1 renders; every collective completes
inside the timed region.  Inputs (scene in HBM, pass buffers allocated) are resident before the
timed region; the timed region contains the render kernels, the resolve kernel and the collective.

A ray = one closest-hit query = one call of bvh4_hit at the reference's render.cuh:47; the
kernels count them on the device.

Prints ONE JSON line on rank 0.  At N = 1 on the headline scene the line also carries, under
"configs", the other BASELINE configurations (orthographic, obj, materials, materials + HDR
environment, 1 M triangles) measured a few frames each after the timed headline region.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Reference-algorithm work per ray: what the REFERENCE's traversal (bvh.cuh:380-421, unordered
# stack walk) executes per closest-hit query, counted with the CPU oracle at reduced image size:
#   flops/ray F = 24*N_box + 84*N_prim + 60*N_accept + 100
#   (box test 21 flop + 3 reciprocals, primitive test 39 + 45, accepted hit 60, scatter/RNG/accumulate 100)
# It is the yardstick for "useful work", not a count of the instructions these kernels issue
# (they test fewer boxes: own hierarchy, nearest child first).
#   scene: (N_pop, N_box, N_prim, N_accept) per ray
REFERENCE_COUNTS = {
    "orthographic.yaml": (2.809, 11.236, 1.018, 0.497),
    "cornell_box.yaml": (7.743, 30.973, 2.527, 0.997),
    "obj.yaml": (12.393, 49.572, 2.100, 0.663),
    "materials.yaml": (4.193, 16.773, 2.209, 0.974),
    "materials_hdr.yaml": (4.193, 16.773, 2.209, 0.974),
    "mesh1m.yaml": (36.837, 147.349, 2.955, 0.832),
    "hdr_env.yaml": (1.000, 4.000, 0.810, 0.296),
    "normal_map_without.yaml": (1.000, 4.000, 1.048, 0.524),
}
STATE_BYTES_PER_RAY = 128.0  # path state + hit record a multi-kernel wavefront design would move per ray
PEAK_FP32_TFLOPS = 157.3     # MI355X vector FP32 peak (packed FMA)
PEAK_HBM_GBS = 8000.0        # MI355X HBM3E peak
HEADLINE = "cornell_box.yaml"


def flops_per_ray(scene):
    c = REFERENCE_COUNTS.get(scene)
    return None if c is None else 24.0 * c[1] + 84.0 * c[2] + 60.0 * c[3] + 100.0


TRAFFIC_CAVEAT = ("an estimate: FETCH_SIZE counts a 128-byte request as 64 bytes on gfx950, hence the factor 2, which is "
                  "calibrated on 16-byte-per-lane streaming reads only and uncalibrated for gather-type accesses")


def measured_traffic(scene):
    """(HBM bytes per launch of the dominant kernel, how they were obtained) from the committed PMC
    passes (separate rocprofv3 --pmc runs of FETCH_SIZE and WRITE_SIZE); (None, None) if this scene
    was not measured."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            rec = json.load(f)[scene]
        return rec["bytes_per_launch"], rec["source"] + "; " + TRAFFIC_CAVEAT
    except (OSError, KeyError, ValueError):
        return None, None


def ensure_assets(scene):
    """mesh1m.yaml reads a generated 1 002 528-triangle height field that is not kept in git."""
    if scene == "mesh1m.yaml":
        obj = os.path.join(ROOT, "scenes", "assets", "objects", "grid708.obj")
        if not os.path.exists(obj):
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_mesh.py"), "grid", "708", obj])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def oracle_rate(scene_file, width, height, spp, bounces, threads):
    """Oracle (CPU restatement of the reference) on `spp` samples of the scene at width x height."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import cudaraytracer_amd as crt
    lib = oracle_lib.load()
    sc = crt.load_scene(scene_file, base_dir=os.path.join(ROOT, "scenes"))
    sc.set_render(width, height, samples=spp, max_bounces=bounces)
    dp = crt.lib().crt_host_scene_desc(sc._host)
    t0 = time.perf_counter()
    _, stats = oracle_lib.render(lib, dp, width, height, samples=spp, threads=threads)
    dt = time.perf_counter() - t0
    sc.close()
    return stats["rays"] / dt / 1e6, stats["rays"], dt


def cpu_baseline(scene_file, width, height, spp_all, threads_all):
    """The oracle timed on this machine's host cores on a bounded sample of the same workload (the
    full-size image at a reduced sample count): one thread and `threads_all` OpenMP threads, plus
    the reference's own CPU-sized case (orthographic 256x256, 8 spp, 10 bounces)."""
    name = os.path.basename(scene_file)
    spp_1 = max(1, min(4, spp_all))
    v1, rays1, dt1 = oracle_rate(scene_file, width, height, spp_1, 0, 1)
    va, raysa, dta = oracle_rate(scene_file, width, height, spp_all, 0, threads_all)
    ortho = os.path.join(ROOT, "scenes", "orthographic.yaml")
    o1, orays, odt1 = oracle_rate(ortho, 256, 256, 8, 10, 1)
    oa, _, odta = oracle_rate(ortho, 256, 256, 8, 10, threads_all)
    return {"value": round(va, 3), "unit": "Mrays/s", "cores": threads_all, "kind": "port",
            "threads_1": round(v1, 3), "threads_all": round(va, 3),
            "host_cores": os.cpu_count(), "cpu_model": cpu_model(),
            "sample": f"{name} {width}x{height}, all bounces: 1 thread on {spp_1} of the spp ({rays1} rays in {dt1:.2f} s), "
                      f"{threads_all} OpenMP threads (this job's CPU share of the node's {os.cpu_count()} logical CPUs) on "
                      f"{spp_all} of the spp ({raysa} rays in {dta:.2f} s)",
            "config1_orthographic_256x256_8spp_10bounces": {"threads_1": round(o1, 3), "threads_all": round(oa, 3), "rays": orays,
                                                            "seconds_1": round(odt1, 3), "seconds_all": round(odta, 3)}}


def roofline_block(scene, rays_per_launch, kernel_ms, resolve_ms, with_traffic):
    """The `roofline` object of the result line for one scene's dominant kernel (render_fused)."""
    flops = flops_per_ray(scene)
    if not flops or kernel_ms <= 0.0:
        return None
    counts = REFERENCE_COUNTS[scene]
    achieved = rays_per_launch * flops / (kernel_ms * 1e-3) / 1e12
    traffic, source = measured_traffic(scene) if with_traffic else (None, None)
    hbm_achieved = None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9
    return {
        "bound": "valu", "kernel": "render_fused", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
        "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
        "kernel_ms": round(kernel_ms, 4), "resolve_ms": round(resolve_ms, 4),
        "rays_per_launch": int(rays_per_launch),
        "flops_per_ray": round(flops, 1),
        "flops_model": "reference-algorithm flops: 24*N_box + 84*N_prim + 60*N_accept + 100 with the counts of the "
                       "reference's own traversal (oracle, reduced image size), not the instructions these kernels issue",
        "reference_counts_per_ray": {"node_pops": counts[0], "box_tests": counts[1], "prim_tests": counts[2], "accepts": counts[3]},
        "hbm": {"achieved": None if hbm_achieved is None else round(hbm_achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": None if hbm_achieved is None else round(hbm_achieved / PEAK_HBM_GBS, 5),
                "source": source,
                "modelled_wavefront_equivalent": {
                    "bytes_per_ray": STATE_BYTES_PER_RAY,
                    "GB/s": round(rays_per_launch * STATE_BYTES_PER_RAY / (kernel_ms * 1e-3) / 1e9, 1),
                    "note": "state a multi-kernel wavefront tracer would move through HBM per ray; the fused kernel keeps it in registers",
                    "scene_bytes_per_ray_reference_walk": round(128.0 * counts[0] + 48.0 * counts[2], 1)}}}


# The BASELINE configurations besides the headline one: (scene file, timed frames, warmup frames).
# (six warmup frames: a scene's first six launches alternate between two orders of its work items, timed by the
# library, before it settles on the faster one)
SETTLE_FRAMES = 6
OTHER_CONFIGS = (("orthographic.yaml", 10, SETTLE_FRAMES), ("obj.yaml", 5, SETTLE_FRAMES), ("materials.yaml", 5, SETTLE_FRAMES),
                 ("materials_hdr.yaml", 5, SETTLE_FRAMES), ("mesh1m.yaml", 2, SETTLE_FRAMES))


def measure_config(crt, torch, scene_name, steps, warmup):
    """One BASELINE configuration on the current GPU, one frame at a time: whole-step Mrays/s over
    `steps` frames (wall clock around launch + kernel + resolve, inputs resident), the render
    kernel's mean HIP-event duration and the roofline fractions that follow from it."""
    ensure_assets(scene_name)
    scene_file = os.path.join(ROOT, "scenes", scene_name)
    sc = crt.load_scene(scene_file, base_dir=os.path.join(ROOT, "scenes"))
    ri = sc.render_info
    W, H, spp, bounces = ri.image_width, ri.image_height, ri.samples_per_pixel, ri.max_bounces
    sc.upload()
    buf = crt.create_buffer(sc)
    lib = crt.lib()

    def frames(n):
        rays, trace_ms, resolve_ms = 0, 0.0, 0.0
        for _ in range(n):
            crt.render_launch(sc, buf, 0, spp)
            _, r = crt.render_finish(sc)
            tr, rs = C.c_float(0), C.c_float(0)
            lib.crt_render_timings(sc.get_device_scene(), C.byref(tr), C.byref(rs))
            rays += r
            trace_ms += tr.value
            resolve_ms += rs.value
        return rays, trace_ms, resolve_ms

    frames(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays, trace_ms, resolve_ms = frames(steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    describe = sc.describe()
    buf.close()
    sc.close()
    roof = roofline_block(scene_name, rays / steps, trace_ms / steps, resolve_ms / steps, True)
    return {"workload": f"{scene_name} {W}x{H}, {spp} spp, {bounces} bounces", "value": round(rays / elapsed / 1e6, 1),
            "unit": "Mrays/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "rays_per_step": rays // steps, "kernel_ms": round(trace_ms / steps, 4), "resolve_ms": round(resolve_ms / steps, 4),
            "frac": None if roof is None else roof["frac"], "achieved_tflops": None if roof is None else roof["achieved"],
            "flops_per_ray": None if roof is None else roof["flops_per_ray"],
            "hbm": None if roof is None else {k: roof["hbm"][k] for k in ("achieved", "peak", "unit", "frac")},
            "traffic": None if roof is None else roof["traffic"], "kernel_variant": describe}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    """Devices this process could use.  torch.cuda.device_count() reads the driver's device list
    without creating a HIP context, so the parent stays free to start children afterwards."""
    import torch
    return torch.cuda.device_count()


def launch_ranks(n, argv, script=None, device_count=None, timeout=None):
    """The parent of an N-rank run: checks that `n` devices are visible, starts the ranks as
    children under torch.distributed.run (fresh processes: nothing here has touched a GPU), relays
    rank 0's JSON line to stdout and returns the exit status (non-zero if a rank failed, printed no
    result, or the devices are missing).  `script` / `device_count` exist for tests: a stand-in
    child program and a stand-in for the number of visible devices."""
    have = visible_gpus() if device_count is None else device_count
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have}; nothing was measured", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        done = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {n} ranks did not finish within {timeout} s", file=sys.stderr)
        return 3
    lines = [ln for ln in done.stdout.decode("utf-8", "replace").splitlines() if ln.lstrip().startswith("{")]
    if done.returncode != 0:
        print(f"bench.py: a rank failed (launcher exit status {done.returncode}); no result reported", file=sys.stderr)
        return done.returncode
    if not lines:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 4
    print(lines[-1], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default=HEADLINE)
    ap.add_argument("--spp", type=int, default=0, help="override the scene's samples per pixel (0 = keep)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = the scene's frame divided over the GPUs (default), weak = the scene's spp per GPU")
    ap.add_argument("--sharding", choices=("auto", "samples", "tiles"), default="auto", help="how a strong-scaling frame is divided")
    ap.add_argument("--cpu-spp", type=int, default=None,
                    help="samples per pixel of the CPU baseline's all-threads leg (0 = skip; default 50 on the headline scene, 0 on any other: "
                         "the CPU restatement needs hours for a frame of the large scenes)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-configs", action="store_true", help="N = 1, headline scene: skip the other BASELINE configurations")
    ap.add_argument("--check", action="store_true",
                    help="N > 1, strong scaling: after the timed region rank 0 renders the whole frame alone and compares it with the "
                         "image the ranks produced together (largest difference in the line; non-zero exit beyond 1e-5 relative)")
    ap.add_argument("--launch", action="store_true",
                    help="start the ranks as child processes even for one GPU (rehearses the parent / child path and the collectives on a one-GPU machine)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.cpu_spp is None:
        args.cpu_spp = 50 if args.scene == HEADLINE else 0

    if os.environ.get("CRT_BENCH_TRACE_AFTER"):
        # debugging aid: every process prints where it stands after that many seconds and exits
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CRT_BENCH_TRACE_AFTER"]), exit=True)

    launched = "RANK" in os.environ
    if not launched and (args.gpus > 1 or args.launch):
        # the parent of an N-rank run: it never touches a GPU itself
        sys.exit(launch_ranks(args.gpus, [a for a in sys.argv[1:] if a != "--launch"]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # every rank sees the same two numbers and leaves together, before any collective
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; nothing was measured")

    if launched:
        # A rank keeps three renders in flight on three streams beside torch's stream and RCCL's: HIP spreads a
        # process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default), and two streams that share one
        # run one after the other - a collective would wait behind a whole render.  Read when HIP starts, so set
        # before anything touches the GPU.
        if int(os.environ.get("GPU_MAX_HW_QUEUES") or 0) < 8:
            os.environ["GPU_MAX_HW_QUEUES"] = "8"

    # stdout carries exactly one line, the result; anything native libraries print on the way
    # (RCCL's version banner, for one) goes to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import cudaraytracer_amd as crt
    from cudaraytracer_amd import multi_gpu

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    # CRT_BENCH_SHARED_GPU=1 (a rehearsal, never a measurement): the ranks share the visible GPUs and
    # talk over gloo (RCCL takes one rank per device), so that a one-GPU machine can walk through the
    # N-rank control flow: the split, the rotation of the buffer sets, the order of the collectives
    shared_gpu = os.environ.get("CRT_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank %= torch.cuda.device_count()
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    crt.set_device(local_rank)
    # under torch.distributed.run (RANK is set) the collective path runs even with one rank, so a
    # single-GPU box can rehearse exactly what the N-GPU launch does
    distributed = launched
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if rank == 0:
        ensure_assets(args.scene)
    if distributed:
        dist.barrier()
    scene_file = os.path.join(ROOT, "scenes", args.scene)
    sc = crt.load_scene(scene_file, base_dir=os.path.join(ROOT, "scenes"))
    if args.spp > 0:
        sc.set_render(samples=args.spp)
    ri = sc.render_info
    W, H, spp, bounces = ri.image_width, ri.image_height, ri.samples_per_pixel, ri.max_bounces
    sc.upload()
    lib = crt.lib()

    # ---- how the frame is divided.  Whether a split gives every rank work depends only on the
    # frame and the world size, so every rank decides alike and they all stop together if it does
    # not (a rank that left on its own would leave the others waiting in the first collective).
    single_gpu_ms = None
    predicted = None
    if not distributed:
        sharding, first, count, total_spp = "single", 0, spp, spp
    elif args.scaling == "weak":
        sharding, total_spp = "samples", spp * world
        first, count = multi_gpu.rank_sample_range(rank, world, spp)
    else:
        total_spp = spp
        try:
            if args.sharding != "auto":
                multi_gpu.check_sharding(args.sharding, H, spp, world)
                sharding = args.sharding
            else:
                # (also with a single rank under a launcher: a one-GPU machine then rehearses every step of the N-rank run)
                multi_gpu.choose_sharding(W, H, spp, world)      # raises if neither split is possible
                # the cost model wants this scene's frame time on one GPU: render the whole frame once
                # on every rank (after one untimed frame) and take the slowest
                probe = crt.create_buffer(sc)
                crt.render(sc, probe)
                t = torch.tensor([crt.render(sc, probe) * 1e3], dtype=torch.float64, device="cpu" if shared_gpu else "cuda")
                probe.close()
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                single_gpu_ms = float(t.item())
                sharding = multi_gpu.choose_sharding(W, H, spp, world, single_gpu_ms)
                predicted = {k: round(multi_gpu.predicted_frame_ms(k, single_gpu_ms, W, H, spp, world), 3)
                             for k in ("tiles", "samples") if multi_gpu.sharding_is_feasible(k, H, spp, world)}
        except ValueError as e:
            raise SystemExit(f"bench.py: {e}; nothing was measured")
        first, count = multi_gpu.split_samples(spp, world)[rank] if sharding == "samples" else (0, spp)

    # Frames in flight: with more than one process the collective of frame k runs while frames k+1 ... render, so
    # `copies` + 2 sets of buffers rotate (a frame's buffers are in use from its launch to the end of its collective)
    # and are rendered into again only after their collective has finished.
    # A rank's share of the frame is short, and a render ends on a few long paths in nearly empty waves: consecutive
    # frames therefore render side by side on three device copies of the scene (own stream and scratch each), every
    # one on a third of the chip (scene option chip_parts), so that one frame's tail runs beside the others' bulk.
    # One rank's eighth of the Cornell frame, frames per time on one GPU (tools/pipeline_blocks.py): 1.10 ms one frame
    # at a time, 0.98 with three full-size renders in flight, 0.87 with three on a third of the chip each; a quarter
    # 1.91 / 1.75 / 1.68, a half 3.58 / 3.39 / 3.29.  A single-GPU run keeps one frame at a time on the whole chip, so
    # that the kernel durations behind the roofline are those of a kernel that has the GPU to itself.
    # (a scene whose tables go far beyond the 4 MB of an XCD's L2 - a million triangles - keeps one frame at a time: every
    # copy is walked through the caches on its own, and three working sets side by side only evict each other)
    copies = 1 if not distributed else max(1, int(os.environ.get("CRT_BENCH_COPIES", "3" if sc.desc.object_count <= 65536 else "1")))
    # (two more buffer sets than scene copies: the collective of frame j starts when frame j has rendered, `copies`
    # iterations after its launch; with one spare set the host would wait for it in the very next iteration, before it
    # launches the next frame, and a third of the chip would idle meanwhile)
    depth = copies + 2 if distributed else 1
    scenes = [sc]
    if copies > 1:
        # (the first copy was uploaded for the whole chip - the sharding probe above renders on it alone - and is uploaded anew)
        sc.set_options(chip_parts=copies)
        sc.upload()
    for _ in range(copies - 1):
        other = crt.load_scene(scene_file, base_dir=os.path.join(ROOT, "scenes"))
        other.set_options(chip_parts=copies)
        if args.spp > 0:
            other.set_render(samples=args.spp)
        other.upload()
        scenes.append(other)
    band_rows = multi_gpu.BAND_ROWS
    if sharding == "tiles":
        own_rows = crt.band_row_count(H, band_rows, rank, world)
        stride = multi_gpu.band_stride_floats(W, H, world)
        own = [torch.zeros(stride, dtype=torch.float32, device="cuda") for _ in range(depth)]
        gathered = [torch.zeros(world * stride, dtype=torch.float32, device="cuda") for _ in range(depth)]
        image = torch.zeros(14 * W * H, dtype=torch.float32, device="cuda")
        bufs = [crt.Buffer(W, own_rows, ri.gamma, device_ptr=t.data_ptr()) for t in own]
        image_buf = crt.Buffer(W, H, ri.gamma, device_ptr=image.data_ptr())
    else:
        passes = [torch.zeros(14 * W * H, dtype=torch.float32, device="cuda") for _ in range(depth)]
        bufs = [crt.Buffer(W, H, ri.gamma, device_ptr=p.data_ptr()) for p in passes]
    # the renders run on the scene's own stream: the fills above (torch's stream) must have landed
    torch.cuda.synchronize()
    pending = [None] * depth

    class HostStagedCollective:
        """Shared-GPU rehearsal only: the collective over gloo on host copies, carried out where the
        caller waits for it (every rank waits for its collectives in the same order)."""

        def __init__(self, slot):
            self.slot = slot

        def wait(self):
            if sharding == "tiles":
                mine, everyone = own[self.slot].cpu(), torch.empty(world * stride, dtype=torch.float32)
                dist.all_gather_into_tensor(everyone, mine)
                gathered[self.slot].copy_(everyone)
            else:
                mean = passes[self.slot].cpu()
                multi_gpu.combine_means(mean, count, total_spp, dist)
                passes[self.slot].copy_(mean)

    def start_collective(slot):
        if not distributed:
            return
        if shared_gpu:
            pending[slot] = HostStagedCollective(slot)
        elif sharding == "tiles":
            pending[slot] = multi_gpu.gather_bands_async(own[slot], gathered[slot], dist)
        else:
            pending[slot] = multi_gpu.combine_means_async(passes[slot], count, total_spp, dist)

    done_events = [torch.cuda.Event() for _ in range(depth)] if distributed else []

    def wait_for(slot):
        """The collective that used this slot's buffers has finished (the host waits for it: the next render into
        them runs on a scene's own stream); the copy kernel that lays the gathered bands out as the image is only
        queued - with the chip full of render kernels it runs when one of them ends, and nothing waits for it here."""
        if pending[slot] is not None:
            pending[slot].wait()
            if not shared_gpu:
                done_events[slot].record()
            if sharding == "tiles":
                crt.assemble_bands(image_buf, gathered[slot].data_ptr(), stride, band_rows, world,
                                   stream=torch.cuda.current_stream().cuda_stream)
            if shared_gpu:
                torch.cuda.current_stream().synchronize()
            else:
                done_events[slot].synchronize()
            pending[slot] = None

    def run(steps):
        """`steps` frames; returns (rays of this rank, summed kernel ms, summed resolve ms)."""
        rays = 0
        trace_ms = resolve_ms = 0.0

        def finish(k):
            nonlocal rays, trace_ms, resolve_ms
            scene = scenes[k % len(scenes)]
            _, r = crt.render_finish(scene)
            tr, rs = C.c_float(0), C.c_float(0)
            lib.crt_render_timings(scene.get_device_scene(), C.byref(tr), C.byref(rs))
            rays += r
            trace_ms += tr.value
            resolve_ms += rs.value
            start_collective(k % depth)

        in_flight = []                                   # frames launched and not yet finished, oldest first
        spent = [0.0, 0.0, 0.0]
        for k in range(steps):
            slot = k % depth
            t_a = time.perf_counter()
            if len(in_flight) == len(scenes):            # this frame's scene copy is still rendering frame k - len(scenes)
                finish(in_flight.pop(0))                 # (its collective is queued before the next render kernel is)
            t_b = time.perf_counter()
            wait_for(slot)                               # the collective of frame k - depth, started two iterations ago
            t_c = time.perf_counter()
            scene = scenes[k % len(scenes)]
            if sharding == "tiles":
                crt.render_bands_launch(scene, bufs[slot], first, count, band_rows, rank, world)
            else:
                crt.render_launch(scene, bufs[slot], first, count)
            t_d = time.perf_counter()
            spent[0] += t_b - t_a; spent[1] += t_c - t_b; spent[2] += t_d - t_c
            in_flight.append(k)
            if len(scenes) == 1:
                finish(in_flight.pop(0))
        while in_flight:
            finish(in_flight.pop(0))
        for slot in range(depth):
            wait_for(slot)
        if os.environ.get("CRT_BENCH_PHASES") and steps:
            print(f"rank {rank}: {steps} frames, host ms per frame in finish+start_collective {spent[0] / steps * 1e3:.3f}, "
                  f"wait_for {spent[1] / steps * 1e3:.3f}, launch {spent[2] / steps * 1e3:.3f}", file=sys.stderr)
        return rays, trace_ms, resolve_ms

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: the frames in which every scene copy settles the order of its work items, then the W warmup steps
    settle = SETTLE_FRAMES * len(scenes)
    run(settle)
    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    rays_rank, trace_ms, resolve_ms = run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_variant = sc.describe()

    check = None
    if args.check and distributed and args.scaling == "strong" and rank == 0:
        # the frame the ranks produced together against the same frame rendered here alone (outside the timed region)
        import numpy as np
        together = (image if sharding == "tiles" else passes[(args.steps - 1) % depth]).cpu().numpy()
        alone = crt.create_buffer(sc)
        alone_seconds = crt.render(sc, alone)
        alone_rays = crt.api.last_ray_count
        host = alone.download()
        alone.close()
        flat = np.concatenate([host[k].ravel() for k in ("color", "albedo", "emission", "normal", "depth", "opacity")])
        both = np.isfinite(flat) & np.isfinite(together)
        # (the N-rank frame groups a pixel's samples differently - per rank, per band - so sums differ in their last bits)
        diff = np.abs(flat[both] - together[both])
        check = {"max_abs_difference": float(diff.max()) if both.any() else 0.0,
                 "max_relative_difference": float((diff / np.maximum(1.0, np.abs(flat[both]))).max()) if both.any() else 0.0,
                 "bit_equal_fraction": float(np.mean(flat.view(np.uint32) == together.view(np.uint32))),
                 "non_finite_agree": bool(np.array_equal(np.isfinite(flat), np.isfinite(together)))}
        # which image rows differ by more than rounding (of the colour pass), and whose bands they are under the tile split
        colour = np.abs(flat[:3 * W * H] - together[:3 * W * H]).reshape(H, W * 3).max(axis=1)
        bad_rows = np.nonzero(~(colour <= 1e-4))[0]
        check["rows_off"] = int(bad_rows.size)
        if bad_rows.size:
            check["rows_off_first"] = [int(r) for r in bad_rows[:8]]
            check["ranks_of_rows_off"] = sorted({int((r // band_rows) % world) for r in bad_rows})
            off = bad_rows[len(bad_rows) // 2]
            check["colour_of_a_row_off"] = {"row": int(off), "alone_mean": float(np.abs(flat[:3 * W * H].reshape(H, W * 3)[off]).mean()),
                                            "together_mean": float(np.abs(together[:3 * W * H].reshape(H, W * 3)[off]).mean())}
            check["scene_after_the_single_render"] = sc.describe()
            check["single_render"] = {"seconds": alone_seconds, "rays": int(alone_rays), "nonzero_fraction": float(np.mean(flat != 0.0))}
            check["together_nonzero_fraction"] = float(np.mean(together != 0.0))

    if distributed:
        where = "cpu" if shared_gpu else "cuda"
        t = torch.tensor([elapsed], dtype=torch.float64, device=where)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        rr = torch.tensor([rays_rank], dtype=torch.int64, device=where)
        dist.all_reduce(rr)
        rays_total = int(rr.item())
    else:
        rays_total = rays_rank

    for b in bufs:
        b.close()
    for scene in scenes:
        scene.close()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = rays_total / elapsed / 1e6
        roofline = roofline_block(args.scene, rays_rank / args.steps, trace_ms / args.steps, resolve_ms / args.steps,
                                  world == 1 and args.spp == 0)
        if roofline is not None and len(scenes) > 1:
            roofline["frames_in_flight"] = len(scenes)
            roofline["note"] = (f"{len(scenes)} renders run side by side, each on 1/{len(scenes)} of the chip: kernel_ms is the duration of a kernel "
                                f"that has a part of the GPU, and achieved / frac follow from it; the single-GPU line has the kernel alone on the chip")
        if not distributed:
            how = "single GPU"
        elif sharding == "tiles":
            how = (f"image tiles: bands of {band_rows} rows round-robin over {world} ranks, all samples per rank, one all-gather of "
                   f"14*W*H floats + assemble kernel per frame, overlapped with the next frame")
        else:
            how = (f"sample ranges per rank ({args.scaling}), one all-reduce of 14*W*H floats per frame, overlapped with the next frame")
        config = {"workload": f"{args.scene} {W}x{H}, {total_spp} spp in total, {bounces} bounces",
                  "spp_this_rank": count, "sharding": how, "rays_per_step": rays_total // args.steps,
                  "settle_frames": settle, "kernel_variant": kernel_variant}
        if shared_gpu:
            config["rehearsal"] = "the ranks share the visible GPUs and talk over gloo: control flow only, not a measurement"
        if check is not None:
            config["check_against_single_render"] = check
        if single_gpu_ms is not None:
            config["sharding_choice"] = {"single_gpu_frame_ms_measured": round(single_gpu_ms, 3), "predicted_frame_ms": predicted,
                                         "chosen": sharding}
        out = {
            "metric": "Mrays/s (whole node) Cornell box 1000x1000, 50spp, 15 bounces" if args.scene == HEADLINE and args.spp == 0
                      else f"Mrays/s (whole node) {args.scene}",
            "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak" if (distributed and args.scaling == "weak") else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": config,
            "roofline": roofline,
        }
        if world == 1 and not distributed and args.scene == HEADLINE and args.spp == 0 and not args.no_configs:
            # the other BASELINE configurations, a few frames each, after the timed headline region
            out["configs"] = []
            for name, steps, warmup in OTHER_CONFIGS:
                try:
                    out["configs"].append(measure_config(crt, torch, name, steps, warmup))
                except Exception as e:          # a configuration that cannot run is reported, not hidden
                    out["configs"].append({"workload": name, "error": str(e)})
        if world == 1 and args.cpu_spp > 0:
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(scene_file, W, H, args.cpu_spp, threads)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if distributed:
        dist.destroy_process_group()
    if check is not None and (check["max_relative_difference"] > 1e-5 or not check["non_finite_agree"]):
        raise SystemExit(f"bench.py: the image of {world} ranks differs from the single render: {check}")


if __name__ == "__main__":
    main()
