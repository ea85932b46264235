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

Full frames sum only the UNION of the ranks' non-zero 4x4x4-voxel bricks (``cpm_allreduce_grid_sparse``: a byte-mask
max-reduce, then a packed payload whose size the host fixed beforehand from the union of two frames ago -- no stream
synchronisation, no read-back on the frame's path; dense after an overflow).  ``sparse_capacity`` is that policy, the same
function the library exports; ``TorchTransport`` carries out the same steps with torch ops for the CPU tests.

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


def sparse_capacity(n_bricks: int, previous_union: int) -> int:
    """Bricks of payload for a union that had `previous_union` bricks two frames ago (< 0: unknown): + 25 % + 64, rounded up to
    64; a quarter of the bricks while unknown; `n_bricks` (= dense) beyond half of them.  Mirrors
    cpm_sparse_reduce_capacity_for (include/cpm/cpm.h) -- tests/test_abi.py holds the two together."""
    if n_bricks <= 0:
        return 0
    if previous_union < 0:
        c = (n_bricks // 4 + 63) & ~63
    else:
        c = (previous_union + previous_union // 4 + 64 + 63) & ~63
    return n_bricks if c * 2 > n_bricks else c


def bricklist_capacity(n_bricks: int, previous_count: int) -> int:
    """Bricks a rank's segment has room for when it listed `previous_count` bricks two frames ago (< 0: unknown): + 25 % + 64 rounded
    up to 64, a quarter of the bricks while unknown, never more than all (rounded up to 64).  Mirrors cpm_bricklist_capacity_for."""
    if n_bricks <= 0:
        return 0
    everything = (n_bricks + 63) & ~63
    c = n_bricks // 4 if previous_count < 0 else previous_count + previous_count // 4 + 64
    c = (c + 63) & ~63
    return 64 if c == 0 else min(c, everything)


def bricklist_segment_bytes(capacity: int, channels: int = 1) -> int:
    """16-byte header + capacity slots of a 16-byte head (the brick's id) and 64 * channels floats (cpm_bricklist_segment_bytes)."""
    return 16 + capacity * (256 * channels + 16)


# What a frame's exchange puts on a rank's busiest xGMI link, and a time for it (arithmetic): the three forms the
# build has, from the brick counts a probe frame measured.  Constants: one link ~153 GB/s peak of which a collective's steady state
# gets about 2/3; a small collective's latency on 8 GPUs ~30 us ("Distributed communication backend").
XGMI_LINK_GBS = 100.0
COLLECTIVE_LATENCY_US = 30.0


def exchange_model(n_bricks: int, channels: int, world: int, union_bricks: int, own_bricks_max: int, cells: int,
                   link_gbs: float = None, latency_us: float = None):
    """Bytes per link and a modelled time per frame of (a) the dense ring reduce, (b) the union-of-bricks reduce
    (cpm_allreduce_grid_sparse with a root: mask max-reduce + packed payload reduce), (c) per-rank brick lists sent to the root
    (cpm_reduce_grid_bricklists: one segment per sender, every sender on its own link into the root).
    union_bricks: bricks non-zero on ANY rank; own_bricks_max: the most any single rank lists.
    link_gbs / latency_us: what one message costs -- the ASSUMED constants above unless the caller measured them (bench.py times a 1 KB
    and a 2 MB send / receive pair over the communicator at set-up: measure_p2p); the figures used are part of the result."""
    brick_bytes = 256 * channels
    ring = (world - 1) / world if world > 1 else 0.0
    dense = cells * channels * 4 * ring
    cap_u = sparse_capacity(n_bricks, union_bricks)
    union = (n_bricks + (cells * channels * 4 if cap_u >= n_bricks else cap_u * brick_bytes)) * ring
    lists = bricklist_segment_bytes(bricklist_capacity(n_bricks, own_bricks_max), channels) if world > 1 else 0
    gbs = float(link_gbs) if link_gbs else XGMI_LINK_GBS
    lat = float(latency_us) if latency_us else COLLECTIVE_LATENCY_US

    def us(nbytes, collectives):
        return collectives * lat + nbytes / (gbs * 1e3)
    return {"dense_reduce": {"bytes_per_link": int(dense), "model_us": round(us(dense, 1), 1)},
            "union_reduce": {"bytes_per_link": int(union), "model_us": round(us(union, 2), 1)},
            "brick_lists": {"bytes_per_link": int(lists), "model_us": round(us(lists, 1), 1)},
            "constants": {"link_gbs": round(gbs, 2), "latency_us": round(lat, 2),
                          "assumed": {"link_gbs": XGMI_LINK_GBS, "latency_us": COLLECTIVE_LATENCY_US},
                          "source": "measured at set-up (measure_p2p)" if (link_gbs or latency_us) else "assumed"}}


def measure_p2p(transport, reps: int = 20):
    """One message's latency and a link's rate over the C-ABI communicator, measured: rank 1 and the root play ping-pong with 1 KB and with
    2 MB messages (cpm_comm_send / cpm_comm_recv on the transport's stream, HIP events around `reps` round trips); every other rank
    stands by.  Returns {"latency_us", "link_gbs", ...} on the two ranks that measured (the caller shares it), None elsewhere or with
    one rank.  A measurement of THIS node's links, not a constant."""
    torch = transport.torch
    if transport.world < 2:
        return None
    root = transport.root if transport.root is not None else 0
    other = 1 if root != 1 else 0
    if transport.rank not in (root, other):
        return None
    peer = other if transport.rank == root else root
    ctx, comm = transport.ctx, transport.comm
    out = {}
    with torch.cuda.stream(transport.stream):
        for name, nbytes in (("small", 1024), ("large", 2 << 20)):
            buf = torch.zeros(nbytes, dtype=torch.uint8, device=ctx.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(reps + 3):
                if it == 3:
                    e0.record(transport.stream)
                if transport.rank == root:
                    ctx.comm_send(comm, buf, nbytes, peer)
                    ctx.comm_recv(comm, buf, nbytes, peer)
                else:
                    ctx.comm_recv(comm, buf, nbytes, peer)
                    ctx.comm_send(comm, buf, nbytes, peer)
            e1.record(transport.stream)
            transport.stream.synchronize()
            out[name] = e0.elapsed_time(e1) * 1e3 / reps / 2.0   # us per one-way message
    lat = out["small"]
    gbs = (2 << 20) / max(out["large"] - lat, 1e-3) / 1e3
    return {"latency_us": round(lat, 2), "link_gbs": round(gbs, 2), "one_way_us": {"1KiB": round(out["small"], 2), "2MiB": round(out["large"], 2)},
            "method": f"ping-pong between ranks {root} and {other}, {reps} round trips per size, HIP events on the reduce stream"}


def scratch_segment(torch, device, n_bricks: int, channels: int = 1, capacity: int = None, ticket: int = 1):
    """A brick-list segment made from plain device buffers (binding.BricklistSegment is the public POD cpm_bricklist_segment): room for
    every brick, `capacity` (default: all) as the exchange's size.  Returns (segment, the tensors that keep it alive).  What tools, tests
    and bench.py's set-up measurement fill without a communicator."""
    from . import binding as B
    room = (n_bricks + 63) & ~63
    buf = torch.empty(16 + room * (16 + 256 * channels), dtype=torch.uint8, device=device)
    ctl = torch.zeros(2, dtype=torch.int32, device=device)
    mail = torch.zeros(1, dtype=torch.int64, device=device)
    seg = B.BricklistSegment(buf.data_ptr(), room if capacity is None else capacity, room, ticket, channels, ctl.data_ptr(), mail.data_ptr())
    return seg, (buf, ctl, mail)


def choose_sender_gather(frame, n_bricks: int, reps: int = 12):
    """How a rank that is NOT the display GPU should produce its brick list -- measured on this rank, with its shard, at set-up (a rank-local
    choice: both forms fill the same segment): "segment" = cpm_gather_fast_segment (the non-zero bricks straight from the gather: no dense
    volume; wins where a rank lights few bricks -- contiguous photon ranges), "pack" = cpm_gather_fast_marked + one pack launch over the marked
    bricks (wins where every rank lights every brick -- tile shards: the segment form pays its slot bookkeeping per gather brick).
    Returns {"chosen", "segment_us", "pack_us"}: the frame's time (trace + bin + gather [+ pack]) in each form."""
    torch, ctx = frame.torch, frame.ctx
    seg, keep = scratch_segment(torch, ctx.device, n_bricks, frame.grid.channels)
    marks = torch.zeros(n_bricks + 16, dtype=torch.uint8, device=ctx.device)
    dense = torch.empty_like(frame.light_volume)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    t_seg = timed(lambda: (frame.trace(), frame.bin_fast(), frame.gather_fast_segment(seg)))
    t_pack = timed(lambda: (frame.trace(), frame.bin_fast(), frame.gather_fast(out=dense, nonzero_bricks=marks),
                            ctx.debug_pack_grid_segment(seg, frame.grid, dense, marks)))
    del keep
    return {"chosen": "segment" if t_seg < t_pack else "pack", "segment_us": round(t_seg, 1), "pack_us": round(t_pack, 1)}


def brick_view(grid, dims, channels=1):
    """The grid (x fastest, `channels` interleaved) as [bz, by, bx, 4, 4, 4 * channels] after zero-padding every axis to a
    multiple of 4: brick b = bx + nbx * (by + nby * bz), a brick's values in (z, y, x, channel) order."""
    import torch
    dx, dy, dz = dims
    px, py, pz = -dx % 4, -dy % 4, -dz % 4
    g = grid.reshape(dz, dy, dx * channels)
    if px or py or pz:
        g = torch.nn.functional.pad(g, (0, px * channels, 0, py, 0, pz))
    nz, ny, nx = (dz + pz) // 4, (dy + py) // 4, (dx + px) // 4
    return g.reshape(nz, 4, ny, 4, nx, 4 * channels).permute(0, 2, 4, 1, 3, 5)


# --------------------------------------------------------------------------- transports

class TorchTransport:
    """all-reduce over torch.distributed (gloo in the CPU tests; any backend).  No-op for one rank.
    root: None = every rank receives the sum (all-reduce); an int = only that rank does (reduce: the display GPU)."""

    def __init__(self, group=None, root=None, p2p_group=None):
        """group: where the collectives run (any backend); p2p_group: where the brick lists' send / recv pairs of host tensors run
        (default: the default group -- gloo in this build's runs)."""
        import torch.distributed as dist
        self._dist, self.group, self.root, self.p2p_group = dist, group, root, p2p_group
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

    # -- the sparse full-frame sum with torch ops: the steps of cpm_allreduce_grid_sparse, one after the other
    def sparse_setup(self, dims, channels=1):
        self._sp = {"dims": tuple(dims), "channels": channels, "unions": [], "log": []}
        self._sp["nb"] = ((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4)

    def sparse_start(self, grid, capacity: int = 0, nonzero_bricks=None):
        """grid (in place) = sum over the ranks, over the union of their non-zero bricks; returns the ticket's figures.
        (nonzero_bricks: accepted for symmetry with RcclTransport; the mask is taken from the grid here.)"""
        import torch
        sp, dist = self._sp, self._dist
        dims, ch, nb = sp["dims"], sp["channels"], sp["nb"]
        k = len(sp["unions"])
        cap = capacity or sparse_capacity(nb, sp["unions"][k - 2] if k >= 2 else -1)
        bricks = brick_view(grid, dims, ch).reshape(nb, 64 * ch).clone()
        mask = (bricks != 0).any(dim=1).to(torch.uint8)
        if self.world > 1:
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        union = torch.nonzero(mask, as_tuple=False).reshape(-1)
        n_union = int(union.numel())
        sp["unions"].append(n_union)
        dense_bytes = grid.numel() * 4
        if cap >= nb or n_union > cap:  # dense by policy, or the union outgrew the payload
            mode = 1 if cap >= nb else 2
            if self.world > 1:
                if self.root is not None:
                    dist.reduce(grid, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.all_reduce(grid, op=dist.ReduceOp.SUM, group=self.group)
            moved = nb + dense_bytes + (cap * 256 * ch if mode == 2 else 0)
        else:
            mode = 0
            payload = torch.zeros(cap, 64 * ch, dtype=grid.dtype, device=grid.device)
            payload[:n_union] = bricks[union]
            if self.world > 1:
                if self.root is not None:
                    dist.reduce(payload, dst=self.root, op=dist.ReduceOp.SUM, group=self.group)
                else:
                    dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=self.group)
            if self.root is None or self._rank() == self.root:
                dx, dy, dz = dims
                bricks[union] = payload[:n_union]  # (`bricks` is a copy: permuted and reshaped)
                nz, ny, nx = (dz + 3) // 4, (dy + 3) // 4, (dx + 3) // 4
                back = bricks.reshape(nz, ny, nx, 4, 4, 4 * ch).permute(0, 3, 1, 4, 2, 5).reshape(nz * 4, ny * 4, nx * 4 * ch)
                grid.reshape(dz, dy, dx * ch).copy_(back[:dz, :dy, :dx * ch])
            moved = nb + cap * 256 * ch
        info = {"n_bricks": nb, "n_union": n_union, "capacity": cap, "mode": mode, "reduce_bytes": moved, "dense_bytes": dense_bytes}
        sp["log"].append(info)
        return info

    def _rank(self):
        return self._dist.get_rank(self.group) if self.world > 1 else 0

    # -- per-rank brick lists to the root with torch ops: the steps of cpm_reduce_grid_bricklists (send / recv pairs; capacities from
    # the counts of two frames before -- a sender's own, at the root every sender's as its headers reported them)
    def lists_setup(self, dims, channels=1, root=0):
        nb = ((dims[0] + 3) // 4) * ((dims[1] + 3) // 4) * ((dims[2] + 3) // 4)
        if nb >= 1 << 24:
            raise ValueError("TorchTransport.lists_setup: this twin carries brick ids as float32 (exact below 2^24 bricks)")
        self._bl = {"dims": tuple(dims), "channels": channels, "nb": nb, "root": root, "counts": [], "log": []}

    def lists_start(self, grid, nonzero_bricks=None):
        """grid at the root (in place) = sum over the ranks; other ranks' grids are read.  Returns the frame's figures (carried out here)."""
        import torch
        bl, dist = self._bl, self._dist
        dims, ch, nb, root = bl["dims"], bl["channels"], bl["nb"], bl["root"]
        rank, k = self._rank(), len(bl["counts"])
        prev = bl["counts"][k - 2] if k >= 2 else None     # per rank (root) / {rank: own} (sender)
        info = {"n_bricks": nb, "n_own": 0, "capacity": 0, "resent": 0, "sent_bytes": 0, "received_bytes": 0, "dense_bytes": grid.numel() * 4, "listed_bricks": 0}
        if self.world == 1:
            bl["counts"].append({})
            bl["log"].append(info)
            return info
        bricks = brick_view(grid, dims, ch).reshape(nb, 64 * ch)

        def segment(values, ids, count, cap):
            seg = torch.zeros(2 + cap + cap * 64 * ch, dtype=torch.float32)
            seg[0], seg[1] = float(count), float(cap)      # (counts stay far below 2^24: exact in float32)
            if count <= cap:
                seg[2:2 + count] = ids.to(torch.float32)
                seg[2 + cap:2 + cap + count * 64 * ch] = values.reshape(-1)
            return seg
        if rank != root:
            mine = torch.nonzero((bricks != 0).any(dim=1), as_tuple=False).reshape(-1)
            # (slots are handed out by a device counter in the library: any order -- here a fixed shuffle)
            mine = mine[torch.randperm(int(mine.numel()), generator=torch.Generator().manual_seed(977 * k + rank)).to(mine.device)]
            count = int(mine.numel())
            cap = bricklist_capacity(nb, prev[rank] if prev is not None else -1)
            vals = bricks[mine].cpu()
            dist.send(segment(vals, mine.cpu(), count, cap), dst=root, group=self.p2p_group)
            info.update(n_own=count, capacity=cap, sent_bytes=bricklist_segment_bytes(cap, ch))
            if count > cap:
                exact = (count + 63) & ~63
                dist.send(segment(vals, mine.cpu(), count, exact), dst=root, group=self.p2p_group)
                info["resent"] = 1
                info["sent_bytes"] += bricklist_segment_bytes(exact, ch)
            bl["counts"].append({rank: count})
        else:
            total = bricks.clone()
            counts = {}
            again = []

            def add(seg, cap, count):
                ids = seg[2:2 + count].to(torch.int64).to(grid.device)
                total[ids] += seg[2 + cap:2 + cap + count * 64 * ch].reshape(count, 64 * ch).to(grid.device)
            for r in range(self.world):     # rank order: a brick several ranks list is summed in that order
                if r == root:
                    continue
                cap = bricklist_capacity(nb, prev[r] if prev is not None else -1)
                seg = torch.zeros(2 + cap + cap * 64 * ch, dtype=torch.float32)
                dist.recv(seg, src=r, group=self.p2p_group)
                info["received_bytes"] += bricklist_segment_bytes(cap, ch)
                count = int(seg[0].item())
                counts[r] = count
                info["listed_bricks"] += count
                if count > cap:
                    again.append((r, count))
                else:
                    add(seg, cap, count)
            for r, count in again:          # a list that had outgrown its segment: again at exact size, added after the others
                cap = (count + 63) & ~63
                seg = torch.zeros(2 + cap + cap * 64 * ch, dtype=torch.float32)
                dist.recv(seg, src=r, group=self.p2p_group)
                info["resent"] += 1
                info["received_bytes"] += bricklist_segment_bytes(cap, ch)
                add(seg, cap, count)
            bl["counts"].append(counts)
            dx, dy, dz = dims
            nz, ny, nx = (dz + 3) // 4, (dy + 3) // 4, (dx + 3) // 4
            back = total.reshape(nz, ny, nx, 4, 4, 4 * ch).permute(0, 3, 1, 4, 2, 5).reshape(nz * 4, ny * 4, nx * 4 * ch)
            grid.reshape(dz, dy, dx * ch).copy_(back[:dz, :dy, :dx * ch])
        bl["log"].append(info)
        return info


class RcclTransport:
    """all-reduce through the C-ABI (cpm_allreduce_grid) on a side stream of the rank's GPU.

    The communicator id is created on rank 0 (cpm_comm_get_unique_id) and handed to the other ranks over
    whatever the host has -- here one torch.distributed broadcast (any backend)."""

    def __init__(self, ctx, rank: int, world: int, group=None, root=None, unique_id: bytes = None):
        import torch
        self.ctx, self.torch, self.world, self.rank, self.root = ctx, torch, world, rank, root
        if unique_id is not None:   # (the host already distributed it: bench.py's probe processes get it on their command line)
            uid = unique_id
        elif world > 1:
            import torch.distributed as dist
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

    def barrier(self):
        """Every rank has reached this call (an 8-float all-reduce on the transport's stream, waited for on the host): the brackets of a
        timed region, without a second communication library on the devices."""
        torch = self.torch
        if self.world > 1:
            if getattr(self, "_token", None) is None:
                self._token = torch.zeros(8, dtype=torch.float32, device=self.ctx.device)
            with torch.cuda.stream(self.stream):
                self.ctx.allreduce_grid(self.comm, self._token)
            self.stream.synchronize()

    # -- the sparse full-frame sum (cpm_allreduce_grid_sparse): same stream discipline as start / wait
    def sparse_setup(self, grid_desc):
        self.sparse = self.ctx.sparse_reduce_create(self.comm, grid_desc)
        self.sparse_log = []

    def sparse_start(self, grid, capacity: int = 0, nonzero_bricks=None):
        """nonzero_bricks: the marks cpm_gather_fast_marked left for this grid (the reduce then needs no pass over it)."""
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            ticket = self.sparse.start(grid, root=-1 if self.root is None else self.root, capacity=capacity, brick_mask=nonzero_bricks,
                                       mask_is_nonzero=nonzero_bricks is not None)
        return ticket

    def sparse_wait(self, ticket):
        """The CURRENT stream waits for the ticket's sum (the dense one after an overflow); the host reads one mailbox word."""
        torch = self.torch
        with torch.cuda.stream(self.stream):
            info = self.sparse.complete(ticket)
            done = torch.cuda.Event()
            done.record(self.stream)
        torch.cuda.current_stream(self.ctx.device).wait_event(done)
        self.sparse_log.append(info)
        return info

    # -- per-rank brick lists to the root (cpm_reduce_grid_bricklists): same stream discipline
    def lists_setup(self, grid_desc, root=0):
        self.lists = self.ctx.bricklist_reduce_create(self.comm, grid_desc, root)
        self.lists_log = []

    def lists_start(self, grid, nonzero_bricks=None):
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            ticket = self.lists.start(grid, nonzero_bricks=nonzero_bricks)
        return ticket

    def lists_open(self):
        """(ticket, segment): the next exchange's ticket and, on a rank other than the root, the segment its gather fills
        (PhotonFrame.gather_fast_segment); segment is None at the root -- it gathers into its grid.  Host only."""
        ticket, seg = self.lists.open()
        return ticket, (seg if seg.segment else None)

    def lists_exchange(self, ticket, root_grid=None, pack_from=None, nonzero_bricks=None):
        """The opened ticket's exchange on the transport's stream, behind everything enqueued on the current one (the gather that filled
        the segment / the root's grid).  pack_from: a sender's dense grid to take the bricks from when its gather did not fill the segment."""
        torch = self.torch
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.ctx.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            if pack_from is not None:
                self.lists.pack_grid(ticket, pack_from, nonzero_bricks)
            self.lists.exchange(ticket, root_grid)
        return ticket

    def lists_wait(self, ticket):
        """The CURRENT stream waits for the ticket's exchange (repeated at exact size where a list had outgrown its segment)."""
        torch = self.torch
        with torch.cuda.stream(self.stream):
            info = self.lists.complete(ticket)
            done = torch.cuda.Event()
            done.record(self.stream)
        torch.cuda.current_stream(self.ctx.device).wait_event(done)
        self.lists_log.append(info)
        return info

    def close(self):
        if getattr(self, "sparse", None) is not None:
            self.sparse.close()
        if getattr(self, "lists", None) is not None:
            self.lists.close()
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

    def __init__(self, like, transport=None, group=None, sparse=None, force=False, lists=None, root=0):
        """sparse: None = dense all-reduce; a cpm GridDesc (RcclTransport) or (dims, channels) (TorchTransport) = the sum
        over the union of the ranks' non-zero bricks (cpm_allreduce_grid_sparse).  lists (same forms; not with sparse): every rank's
        own non-zero bricks as a list to `root`, whose grid becomes the sum (cpm_reduce_grid_bricklists: for shards with disjoint
        brick sets -- contiguous photon ranges).  force: run the reduce with one rank too (measuring pack / unpack on one GPU)."""
        import torch
        if sparse is not None and lists is not None:
            raise ValueError("OverlappedGridReducer: sparse (union of bricks) or lists (per-rank brick lists), not both")
        self.transport = transport if transport is not None else TorchTransport(group)
        self.active = self.transport.world > 1 or force
        self.buffers = [like, torch.empty_like(like)]
        self._pending = [None, None]
        self.sparse = sparse is not None
        self.lists = lists is not None
        self.info = []  # per completed sparse ticket: union, capacity, mode, bytes; per completed list exchange: its cpm_bricklist_info
        self.marks = [None, None]  # per buffer: the non-zero 4x4x4 bricks, written by the gather (marks_for)
        self._opened = {}          # frame -> (ticket, segment) of a brick-list exchange opened by segment_for and not yet enqueued
        self.use_segments = True   # brick lists over the C-ABI: senders gather straight into their segments (False: dense volume + pack launch)
        if (self.sparse or self.lists) and self.active:
            desc = sparse if self.sparse else lists
            if isinstance(self.transport, TorchTransport):
                dims, channels = desc
                if self.sparse:
                    self.transport.sparse_setup(dims, channels)
                else:
                    self.transport.lists_setup(dims, channels, root)
            else:
                if self.sparse:
                    self.transport.sparse_setup(desc)
                    nb = self.transport.sparse.n_bricks
                else:
                    self.transport.lists_setup(desc, root)
                    nb = self.transport.lists.n_bricks
                self.marks = [torch.zeros(nb + 16, dtype=torch.uint8, device=like.device) for _ in range(2)]

    def _wait(self, b):
        h = self._pending[b]
        if h is None:
            return
        self._pending[b] = None
        if not (self.sparse or self.lists):
            self.transport.wait(h)
        elif isinstance(self.transport, TorchTransport):
            self.info.append(h)  # carried out at start
        elif self.lists:
            i = self.transport.lists_wait(h)
            self.info.append({f: int(getattr(i, f)) for f in ("n_bricks", "n_own", "capacity", "resent", "sent_bytes", "received_bytes",
                                                              "dense_bytes", "listed_bricks")})
        else:
            i = self.transport.sparse_wait(h)
            self.info.append({"n_bricks": i.n_bricks, "n_union": i.n_union, "capacity": i.capacity, "mode": i.mode,
                              "reduce_bytes": i.reduce_bytes, "dense_bytes": i.dense_bytes})

    def acquire(self, k: int):
        b = k & 1
        self._wait(b)
        return self.buffers[b]

    def segment_for(self, k: int):
        """Brick lists over the C-ABI, on a rank that is not the root: the segment frame k's gather writes its non-zero bricks into
        (PhotonFrame.gather_fast_segment) -- then no dense buffer, no zeros and no pass over one on this rank; None where the frame
        gathers into acquire(k) as ever (the root, other exchanges, the torch twin).  Call after acquire(k)."""
        if not (self.lists and self.active and self.use_segments) or isinstance(self.transport, TorchTransport):
            return None
        if k not in self._opened:
            self._opened[k] = self.transport.lists_open()
        return self._opened[k][1]

    def marks_for(self, k: int):
        """Where the gather of frame k leaves the non-zero bricks of its volume (cpm_gather_fast_marked), or None: pass it as the
        gather's `nonzero_bricks` and say so to reduce(k, marked=True) -- the reduce then skips its own pass over the volume."""
        return self.marks[k & 1] if self.active else None

    def reduce(self, k: int, marked: bool = False):
        if self.active:
            b = k & 1
            if self.sparse:
                self._pending[b] = self.transport.sparse_start(self.buffers[b], nonzero_bricks=self.marks[b] if marked else None)
            elif self.lists and k in self._opened:
                # (the ticket was opened by segment_for: a sender's gather has filled its segment, the root's its grid)
                ticket, seg = self._opened.pop(k)
                self._pending[b] = self.transport.lists_exchange(ticket, self.buffers[b] if seg is None else None)
            elif self.lists:
                self._pending[b] = self.transport.lists_start(self.buffers[b], nonzero_bricks=self.marks[b] if marked else None)
            else:
                self._pending[b] = self.transport.start(self.buffers[b])

    def flush(self):
        for b in (0, 1):
            self._wait(b)

    def result(self, k: int):
        return self.buffers[k & 1]
