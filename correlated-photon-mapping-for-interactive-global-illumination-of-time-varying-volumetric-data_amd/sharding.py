"""Photon sharding across the GPUs of one node.

Photon i depends only on light sample i, RNG stream i and the read-only volume / TF, so a rank
simply owns a contiguous range of photon indices; every rank bins and gathers its photons into its
own full-size irradiance grid and the grids are summed with ONE collective per frame
(RCCL all-reduce over xGMI; backend "nccl" is RCCL on ROCm).  No other data-path communication.
"""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous photon range [lo, hi) of `rank`; ranges tile [0, n_total) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_light_volume(grid, group=None):
    """Sum the per-rank irradiance grids in place (no-op without an initialised process group)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(grid, op=dist.ReduceOp.SUM, group=group)
    return grid
