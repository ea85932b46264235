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


class OverlappedGridReducer:
    """The frame's one collective, taken off the critical path.

    The all-reduce of frame k's irradiance grid (8 MiB at 128^3: latency-bound on xGMI, of the order of
    the 0.2 ms frame itself) runs asynchronously while the rank already traces and bins frame k + 1,
    which do not touch the grid.  Grids are double-buffered: frame k gathers into buffer k mod 2, and
    before a buffer is gathered into again the reduce that was using it (frame k - 2) is waited for --
    on the stream, not on the host.  `flush()` waits for everything outstanding (end of a timed
    region / before the grid is read).

        red = OverlappedGridReducer(like=frame.light_volume)
        for k in range(K):
            frame.trace(); frame.bin()
            out = red.acquire(k)          # buffer of frame k, safe to overwrite
            frame.gather(out=out)
            red.reduce(k)                 # async all-reduce of that buffer
        red.flush()
        grid = red.result(K - 1)
    """

    def __init__(self, like, group=None):
        import torch
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.buffers = [like, torch.empty_like(like)]
        self._pending = [None, None]

    def acquire(self, k: int):
        b = k & 1
        if self._pending[b] is not None:
            self._pending[b].wait()
            self._pending[b] = None
        return self.buffers[b]

    def reduce(self, k: int):
        if self.active:
            b = k & 1
            self._pending[b] = self._dist.all_reduce(self.buffers[b], op=self._dist.ReduceOp.SUM, group=self.group,
                                                     async_op=True)

    def flush(self):
        for b in (0, 1):
            if self._pending[b] is not None:
                self._pending[b].wait()
                self._pending[b] = None

    def result(self, k: int):
        return self.buffers[k & 1]
