"""Multi-GPU harness for the conv path: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no distributed code at all; this is the sharding SURVEY.md
section 8e defines.  The op has no cross-image term (quantconv2d.cu:83: batch = index / (OC*OH*OW)), so
a global batch shards by image with NO data-path collective; weights, descriptions and scales are
replicated.  The only exchange is the top-1 tail: one all-gather of fp32 logits per step
(per rank (B,1000) = 1 MB at B=256: latency-bound on xGMI, far below a link's 153 GB/s).
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous image range [lo, hi) of `rank`: rank r gets images [256r, 256r+256) when
    total = 256 * world_size; a remainder is spread over the first ranks."""
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(x, world_size=None, rank=None):
    """The slice of a global NCHW batch this rank convolves."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    lo, hi = shard_bounds(x.shape[0], world_size, rank)
    return x[lo:hi]


def gather_logits(logits, group=None, equal_shards=None, force_collective=False):
    """All-gather per-rank (B_r, C) logits into the global (sum B_r, C) tensor, rank order = image
    order.  Equal shards use one all_gather_into_tensor (a single RCCL ncclAllGather); unequal
    shards fall back to a padded gather.

    equal_shards=True is the caller's promise that every rank holds the same number of rows (the bench:
    256 images per rank) and skips the size exchange.  Otherwise the row counts are all-gathered on EVERY
    call, so all ranks always take the same branch: a decision cached per rank (round 1) let two ranks issue
    different collectives when the global batch changed between calls (7 then 8 images on 2 ranks: hang).

    force_collective=True issues the collective even in a group of one (where the gather is the identity): the
    single-GPU RCCL sanity test uses it to prove that the library loads and accepts exactly this call form."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return logits
    logits = logits.contiguous()
    counts = None
    if not equal_shards:
        sizes = torch.tensor([logits.shape[0]], device=logits.device, dtype=torch.int64)
        all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes, group=group)
        counts = [int(s.item()) for s in all_sizes]
    if counts is None or min(counts) == max(counts):
        out = torch.empty((world * logits.shape[0],) + tuple(logits.shape[1:]), dtype=logits.dtype,
                          device=logits.device)
        dist.all_gather_into_tensor(out, logits, group=group)
        return out
    m = max(counts)
    padded = torch.zeros((m,) + tuple(logits.shape[1:]), dtype=logits.dtype, device=logits.device)
    padded[: logits.shape[0]] = logits
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def top1(logits, targets=None):
    """argmax over classes; with targets, the top-1 accuracy in percent (utils/tools.py:63-70)."""
    pred = logits.argmax(dim=1)
    if targets is None:
        return pred
    return pred, float((pred == targets).float().mean().item() * 100.0)
