"""Sharding of one frame over the GPUs of a node (one process per GPU).

The reference renders on one GPU.  Every (pixel, sample number) path is independent given the
keyed RNG, so a frame can be cut two ways:

  samples  rank k renders a range of the sample numbers of every pixel into its own pass buffers
           (as a mean over its range); ONE all-reduce (sum) of the 14*W*H pass floats, each rank's
           mean pre-scaled by its share of the samples, combines them.  NaN depth (a missed
           sample) propagates through the sum exactly like it does through the reference's
           running mean (render.cuh:147-152).
  tiles    the image is cut into bands of BAND_ROWS pixel rows dealt round-robin to the ranks;
           rank k renders ALL samples of its rows into a buffer that holds just those rows, ONE
           all-gather hands every rank the other bands and a copy kernel
           (crt_buffers_assemble_bands) lays them out as the whole image.  No reduction, half
           the bytes of the all-reduce on the wire, and the work divides evenly whatever the
           sample count (50 samples over 8 GPUs leave a 7-against-6.25 imbalance in the samples
           split: 12 % of the frame).

choose_sharding() picks per frame from a cost estimate of both.
"""

BAND_ROWS = 8          # pixel rows per band: one 8x8 tile row (two 8x4 tile rows of the tree walk)

# Cost model constants (per GPU): xGMI gives 7 links x ~153 GB/s peer to peer; a ring collective is
# bound by one link, RCCL reaches roughly half of it on messages of tens of megabytes.
LINK_GBS = 153.0
RING_EFFICIENCY = 0.5
COLLECTIVE_LATENCY_MS = 0.05


def rank_sample_range(rank, world, spp_per_rank):
    """Weak scaling: (first_sample, count), rank k renders sample numbers first+1 .. first+count."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * spp_per_rank, spp_per_rank


def split_samples(total_spp, world):
    """Strong-scaling split of `total_spp` samples: list of (first_sample, count) per rank."""
    base, extra = divmod(total_spp, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((first, n))
        first += n
    return out


def band_rows_of_rank(height, rank, world, band_rows=BAND_ROWS):
    """Pixel rows (ascending) that `rank` owns in the tile split: j with (j // band_rows) % world == rank."""
    return [j for j in range(height) if (j // band_rows) % world == rank]


def predicted_frame_ms(sharding, single_gpu_ms, width, height, spp, world):
    """Estimate of one frame's critical path on `world` GPUs: the slowest rank's render plus the
    collective (which the next frame's render hides when frames are pipelined; a single frame
    pays it)."""
    bytes_total = 14.0 * 4.0 * width * height
    bw = LINK_GBS * RING_EFFICIENCY * 1e6          # bytes per millisecond
    if world == 1:
        return single_gpu_ms
    if sharding == "samples":
        share = max(n for _, n in split_samples(spp, world)) / float(spp)
        comm = 2.0 * (world - 1) / world * bytes_total / bw + COLLECTIVE_LATENCY_MS
    elif sharding == "tiles":
        rows = max(len(band_rows_of_rank(height, r, world)) for r in range(world))
        share = rows / float(height)
        comm = (world - 1) / world * bytes_total / bw + COLLECTIVE_LATENCY_MS
    else:
        raise ValueError(sharding)
    return single_gpu_ms * share + comm


def band_count(height, band_rows=BAND_ROWS):
    """Number of bands an image of `height` rows is cut into."""
    return (height + band_rows - 1) // band_rows


def sharding_is_feasible(sharding, height, spp, world):
    """True when EVERY rank gets work under this split: at least one sample number each for
    'samples', at least one band of rows each for 'tiles'.  A pure function of the frame and the
    world size, so that every rank comes to the same answer without talking to the others."""
    if sharding == "samples":
        return spp >= world
    if sharding == "tiles":
        return band_count(height) >= world
    raise ValueError(sharding)


def check_sharding(sharding, height, spp, world):
    """Raises ValueError (the same one on every rank, before any collective) when `sharding` would
    leave a rank without work: a rank that stopped on its own would leave the others waiting in
    the collective until its timeout."""
    if not sharding_is_feasible(sharding, height, spp, world):
        if sharding == "samples":
            raise ValueError(f"{spp} samples per pixel cannot be divided over {world} GPUs (a rank would get none); "
                             f"use the tile split or fewer GPUs")
        raise ValueError(f"an image of {height} rows has {band_count(height)} bands of {BAND_ROWS} rows, fewer than "
                         f"{world} GPUs (a rank would own no rows); use the sample split or fewer GPUs")


def choose_sharding(width, height, spp, world, single_gpu_ms=10.0):
    """'tiles' or 'samples' for this frame: the cheaper prediction among the splits that give every
    rank work (`single_gpu_ms`: the frame's render time on one GPU, measured by the caller where it
    can be).  Raises ValueError when neither split does (fewer samples AND fewer bands than GPUs)."""
    if world == 1:
        return "samples"
    tiles_ok = sharding_is_feasible("tiles", height, spp, world)
    samples_ok = sharding_is_feasible("samples", height, spp, world)
    if not tiles_ok and not samples_ok:
        raise ValueError(f"neither split gives each of {world} GPUs work: {spp} samples per pixel and "
                         f"{band_count(height)} bands of {BAND_ROWS} rows; use fewer GPUs")
    if not tiles_ok:
        return "samples"
    if not samples_ok:
        return "tiles"
    t = predicted_frame_ms("tiles", single_gpu_ms, width, height, spp, world)
    s = predicted_frame_ms("samples", single_gpu_ms, width, height, spp, world)
    return "tiles" if t <= s else "samples"


def combine_means(passes, count, total, dist=None):
    """In-place: passes (a torch tensor holding this rank's MEAN over `count` samples) becomes the
    mean over all `total` samples of all ranks.  `dist` is torch.distributed (initialised) or None
    for a single rank."""
    if dist is None or not dist.is_initialized():
        return passes
    if count != total:
        passes.mul_(float(count) / float(total))
    dist.all_reduce(passes)
    return passes


def combine_means_async(passes, count, total, dist):
    """combine_means without waiting: returns the collective's work handle (None for a single
    process).  The caller waits on it before the tensor is read or written again."""
    if dist is None or not dist.is_initialized():
        return None
    if count != total:
        passes.mul_(float(count) / float(total))
    return dist.all_reduce(passes, async_op=True)


def band_stride_floats(width, height, world, band_rows=BAND_ROWS):
    """Floats per rank in the all-gather buffer: the largest band's 14 floats per pixel."""
    return 14 * width * max(len(band_rows_of_rank(height, r, world, band_rows)) for r in range(world))


def gather_bands_async(own, gathered, dist):
    """All-gather of the ranks' band buffers.  `own`: this rank's tensor of band_stride_floats()
    elements (its six pass arrays back to back at the start, the rest padding); `gathered`: tensor
    of world * band_stride_floats() elements.  Returns the work handle."""
    return dist.all_gather_into_tensor(gathered, own, async_op=True)


def assemble_bands_reference(gathered, width, height, world, band_rows=BAND_ROWS):
    """What crt_buffers_assemble_bands computes, on a CPU tensor / array (for tests): the 14*W*H
    floats of the whole image in the six-array layout."""
    import numpy as np
    g = np.asarray(gathered, dtype=np.float32).reshape(world, -1)
    n_full = width * height
    out = np.zeros(14 * n_full, np.float32)
    for r in range(world):
        rows = band_rows_of_rank(height, r, world, band_rows)
        n = len(rows) * width
        if not n:
            continue
        rows = np.asarray(rows)
        for k in range(4):                                                   # color albedo emission normal
            src = g[r, 3 * k * n:3 * (k + 1) * n].reshape(len(rows), width, 3)
            out[3 * k * n_full:3 * (k + 1) * n_full].reshape(height, width, 3)[rows] = src
        for k in range(2):                                                   # depth opacity
            src = g[r, (12 + k) * n:(13 + k) * n].reshape(len(rows), width)
            out[(12 + k) * n_full:(13 + k) * n_full].reshape(height, width)[rows] = src
    return out
