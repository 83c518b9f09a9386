"""Multi-GPU sharding of the render path: one process per GPU, scene replicated,
work split by stereo eye, by samples per pixel, or by image rows.

Pixels, samples and eyes are independent in the reference kernel (one thread per
pixel, RayTracingShader.compute:725, no inter-pixel reads in the plain mode), so
the only exchange step is the sum of per-rank accumulation buffers (sample split)
or a gather of disjoint rows (tile split).  `torch.distributed` carries it: the
"nccl" backend is RCCL over xGMI on MI355X, "gloo" on CPU for tests.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_spp(total_spp, world):
    """Samples per rank, differing by at most one, summing to total_spp."""
    base, extra = divmod(int(total_spp), int(world))
    return [base + (1 if r < extra else 0) for r in range(world)]


def sample_index(step, rank, world):
    """Global index of the frame rank `rank` renders in step `step`; it selects the frame's _Seed."""
    return step * world + rank


def tile_rows(height, rank, world):
    """Contiguous block of image rows [y0, y0+h) owned by `rank` (tile split, latency mode)."""
    counts = shard_spp(height, world)
    y0 = sum(counts[:rank])
    return y0, counts[rank]


def eye_for_rank(rank, world):
    """Stereo split: even ranks render the left eye, odd ranks the right (-1 / +1)."""
    return -1 if rank % 2 == 0 else 1


@dataclass
class ShardedAccumulator:
    """Sum of frames over steps and ranks, kept on rank `dst`.

    Each rank hands in the frame it rendered (a mean over its own samples, weight =
    number of samples); frames are weighted, sum-reduced to `dst` in ONE collective
    per call, and added to the running sum there.  image() divides by the total
    weight.  fp32 sums across ranks are order-dependent, so the result matches a
    single-process render to rounding (RMSE ~1e-7 relative), not bit for bit.
    """
    dst: int = 0
    total: torch.Tensor = None
    weight: float = 0.0

    def add(self, frame, samples, group=None, weight=None, reduce=None, all_reduce=None):
        """`frame` times `samples` goes into the sum and `weight` (default: `samples`) into the total weight.  `reduce` /
        `all_reduce` replace the torch.distributed calls (e.g. to stage CUDA tensors through the host on a backend
        that cannot take them)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        weighted = frame * float(samples) if samples != 1 else frame
        w = torch.tensor([float(samples if weight is None else weight)], dtype=torch.float64, device=frame.device)
        # a process group of one rank still runs the collectives (the identity then): that is how a one-GPU box rehearses
        # the very calls an N-GPU job makes
        if world > 1 or dist.is_initialized():
            if weighted is frame:
                weighted = frame.clone()
            (reduce or dist.reduce)(weighted, dst=self.dst, op=dist.ReduceOp.SUM, group=group)
            (all_reduce or dist.all_reduce)(w, op=dist.ReduceOp.SUM, group=group)
        if rank == self.dst:
            if self.total is None:
                self.total = torch.zeros_like(weighted)
            self.total += weighted
        self.weight += float(w.item())

    def image(self):
        if self.total is None:
            return None
        return self.total / self.weight
