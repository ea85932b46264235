"""Photon sharding across the GPUs of one node.

Photon i depends only on light sample i, RNG stream i and the read-only volume / TF, so a rank
simply owns a set of photon indices; every rank bins and gathers its photons into its own full-size
irradiance grid and the grids are summed with ONE collective per frame.  No other data-path
communication.

Which indices: `shard_tiles` (default of bench.py and the drivers) deals the 4096-sample tiles of the
emission lattice round-robin -- rank r takes tiles t = r (mod N), the addressing the reference uses to
put several lights into one buffer (photonOffset + thread, ref progressivephotonmapping/cl/
photontracer.cl:102,123,166) applied per tile -- so every rank sees every lit brick at 1/N of the
density and the one-workgroup-per-brick gather stays balanced.  `shard_range` (a contiguous range = a
slab of the light plane: a rank's photons pile into 1/N of the lit bricks, N times denser; measured
12.9 -> 32 us for the brick gather at N = 8) is kept for comparison.

The collective goes through the C-ABI: ``cpm_allreduce_grid`` (RCCL over xGMI on the caller's stream,
``include/cpm/cpm.h``), the call a C++ host makes.  ``torch.distributed`` only carries the 128-byte
communicator id to the ranks (and the benchmark's barriers).  For the CPU tests (gloo, no GPU, no
RCCL) the same classes run over ``TorchTransport``: identical sharding, buffering and waiting logic,
``dist.all_reduce`` as the wire.
"""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous photon range [lo, hi) of `rank`; ranges tile [0, n_total) exactly."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


SHARD_TILE = 4096  # samples per tile: the tracer's XCD tile (cpm_trace.hip) and the bin's count tile (cpm_fastvolume.hip)


def shard_tiles(n_total: int, rank: int, world: int, tile: int = SHARD_TILE):
    """Global photon indices (ascending, int64 numpy array) of `rank` when the `tile`-sample tiles of the lattice are
    dealt round-robin: tile t belongs to rank t mod world.  The shards partition [0, n_total) and differ by at most
    one tile in size; local photon j of the rank is global photon shard_tiles(...)[j]."""
    import numpy as np
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    if tile < 1:
        raise ValueError("tile must be positive")
    n_tiles = -(-n_total // tile)
    mine = np.arange(rank, n_tiles, world, dtype=np.int64)
    if mine.size == 0:
        return np.zeros(0, np.int64)
    idx = (mine[:, None] * tile + np.arange(tile, dtype=np.int64)[None, :]).reshape(-1)
    return idx[idx < n_total]


# --------------------------------------------------------------------------- transports

class TorchTransport:
    """all-reduce over torch.distributed (gloo in the CPU tests; any backend).  No-op for one rank.
    root: None = every rank receives the sum (all-reduce); an int = only that rank does (reduce: the display GPU)."""

    def __init__(self, group=None, root=None):
        import torch.distributed as dist
        self._dist, self.group, self.root = dist, group, root
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def start(self, grid):
        if self.world > 1:
            if self.root is not None:
                return self._dist.reduce(grid, dst=self.root, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)
            return self._dist.all_reduce(grid, op=self._dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    @staticmethod
    def wait(handle):
        if handle is not None:
            handle.wait()


class RcclTransport:
    """all-reduce through the C-ABI (cpm_allreduce_grid) on a side stream of the rank's GPU.

    The communicator id is created on rank 0 (cpm_comm_get_unique_id) and handed to the other ranks over
    whatever the host has -- here one torch.distributed broadcast (any backend)."""

    def __init__(self, ctx, rank: int, world: int, group=None, root=None):
        import torch
        import torch.distributed as dist
        self.ctx, self.torch, self.world, self.rank, self.root = ctx, torch, world, rank, root
        if world > 1:
            uid = [ctx.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0, group=group)
            uid = uid[0]
        else:
            uid = ctx.comm_unique_id()
        self.comm = ctx.comm_create(uid, rank, world)
        self.stream = torch.cuda.Stream(device=ctx.device)

    def start(self, grid):
        """Enqueue the reduce behind everything already enqueued on the current stream; returns an event."""
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if self.root is None:
                self.ctx.allreduce_grid(self.comm, grid)
            else:  # cpm_reduce_grid: only the root's grid becomes the sum (in place there), the others only send
                self.ctx.reduce_grid(self.comm, grid, grid if self.rank == self.root else None, self.root)
            done = torch.cuda.Event()
            done.record(self.stream)
        return done

    def wait(self, handle):
        """The CURRENT stream waits for the reduce (no host wait)."""
        if handle is not None:
            self.torch.cuda.current_stream(self.ctx.device).wait_event(handle)

    def close(self):
        self.comm.close()


def allreduce_light_volume(grid, transport=None, group=None):
    """Sum the per-rank irradiance grids in place and wait for the result."""
    tr = transport if transport is not None else TorchTransport(group)
    tr.wait(tr.start(grid))
    return grid


class OverlappedGridReducer:
    """The frame's one collective, taken off the critical path.

    The all-reduce of frame k's irradiance grid (8 MiB at 128^3: latency-bound on xGMI, of the order of
    the frame itself) runs asynchronously while the rank already traces and bins frame k + 1, which do
    not touch the grid.  Grids are double-buffered: frame k gathers into buffer k mod 2, and before a
    buffer is gathered into again the reduce that was using it (frame k - 2) is waited for -- on the
    stream, not on the host.  `flush()` waits for everything outstanding (end of a timed region / before
    the grid is read).

        red = OverlappedGridReducer(frame.light_volume, transport)
        for k in range(K):
            frame.trace(); frame.bin()
            out = red.acquire(k)          # buffer of frame k, safe to overwrite
            frame.gather(out=out)
            red.reduce(k)                 # async all-reduce of that buffer
        red.flush()
        grid = red.result(K - 1)
    """

    def __init__(self, like, transport=None, group=None):
        import torch
        self.transport = transport if transport is not None else TorchTransport(group)
        self.active = self.transport.world > 1
        self.buffers = [like, torch.empty_like(like)]
        self._pending = [None, None]

    def acquire(self, k: int):
        b = k & 1
        if self._pending[b] is not None:
            self.transport.wait(self._pending[b])
            self._pending[b] = None
        return self.buffers[b]

    def reduce(self, k: int):
        if self.active:
            b = k & 1
            self._pending[b] = self.transport.start(self.buffers[b])

    def flush(self):
        for b in (0, 1):
            if self._pending[b] is not None:
                self.transport.wait(self._pending[b])
                self._pending[b] = None

    def result(self, k: int):
        return self.buffers[k & 1]
