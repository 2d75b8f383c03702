"""Data-parallel driver: images are independent units, so each rank runs the whole integer forward on
its own shard with replicated weights; the ONLY exchange is one all-gather of the per-image top-1
indices (int32[B/rank]) per batch -- RCCL over xGMI on GPUs (torch.distributed backend "nccl"),
gloo in the CPU tests.  SURVEY.md §8e.  (The reference has no distributed path at all: its NCCL
helpers in /root/reference/utils/utils.py:215-237 are dead code.)"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous shard [lo, hi) of n images for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_top1(local_top1: torch.Tensor, world: int, counts=None) -> torch.Tensor:
    """all-gather of the per-rank top-1 vectors -> [sum of the shard sizes] in rank order.  `counts`: shard size of every rank
    when they differ (shard_bounds of a batch that the world does not divide): every rank pads its vector to the largest shard,
    ONE all_gather_into_tensor as for equal shards, and the padding is cut out of the result."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_top1      # no process group: a plain single-GPU run (a one-rank group still runs the collective)
    if counts is not None and len(set(counts)) > 1:
        assert len(counts) == world and local_top1.numel() == counts[dist.get_rank()]
        width = max(counts)
        padded = torch.zeros(width, dtype=local_top1.dtype, device=local_top1.device)
        padded[: local_top1.numel()] = local_top1
        out = torch.empty(world * width, dtype=local_top1.dtype, device=local_top1.device)
        dist.all_gather_into_tensor(out, padded)
        return torch.cat([out[r * width: r * width + c] for r, c in enumerate(counts)])
    out = torch.empty(world * local_top1.numel(), dtype=local_top1.dtype, device=local_top1.device)
    dist.all_gather_into_tensor(out, local_top1.contiguous())
    return out


class DataParallelTop1:
    """graph=True: the local forward is a HIP-graph replay reading `local_images` in place (the caller refills that tensor
    between steps); the all-gather stays an ordinary stream-ordered RCCL call after it.  counts: see gather_top1."""

    def __init__(self, engine, world: int, graph: bool = False, counts=None):
        self.engine, self.world, self.graph, self.counts = engine, world, graph, counts

    def step(self, local_images: torch.Tensor) -> torch.Tensor:
        if self.graph:
            _, _, top1 = self.engine.forward_graph(local_images, resident=True)
        else:
            _, _, top1 = self.engine(local_images)
        return gather_top1(top1, self.world, self.counts)
