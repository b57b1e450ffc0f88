"""Sample-range sharding of one frame over the GPUs of a node.

The reference renders on one GPU.  Every (pixel, sample number) path is independent given the
keyed RNG, so rank k of N renders sample numbers k*spp+1 .. (k+1)*spp into its own pass buffers
(as a mean over its range) and the ranks combine with ONE collective: an all-reduce (sum) of the
14*W*H pass floats, each rank's mean pre-scaled by its share of the samples.  NaN depth (a
missed sample) propagates through the sum exactly like it does through the reference's running
mean (render.cuh:147-152).
"""


def rank_sample_range(rank, world, spp_per_rank):
    """(first_sample, count): rank k renders sample numbers first+1 .. first+count."""
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
