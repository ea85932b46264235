"""Frame orchestration over the C-ABI: emission -> trace -> bin -> gather.

This is the thin Python driver used by bench.py, smoke() and the parity tests.
It mirrors what the reference's two processors do per evaluation
(ProgressivePhotonTracerCL::process, ref
progressivephotonmapping/processor/progressivephotontracercl.cpp:219-260,541-560 and
PhotonToLightVolumeProcessorCL::process, ref
.../photontolightvolumeprocessorcl.cpp:137-170,299-339,356-412) and nothing else;
the C++ drop-in surface lives in host/.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import binding as B
from . import synthetic as S


# --------------------------------------------------------------------------- E2 / C2 (host arithmetic, CPU in the reference too)
# One implementation: host/cpm_hostmath.cpp (what the C++ processors use), reached here through libcpm_hostmath.so.

def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return v / np.float32(np.sqrt(np.sum(v * v, dtype=np.float32)))


_hostmath_lib = None


def _hostmath():
    global _hostmath_lib
    if _hostmath_lib is None:
        from . import build as _build
        lib = C.CDLL(str(_build.build_hostmath_library(verbose=False)))
        fp, dp, i32 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int
        lib.cpmh_fit_light_rectangle.restype, lib.cpmh_fit_light_rectangle.argtypes = None, [fp, i32, fp, fp, fp]
        lib.cpmh_hull_cycle.restype, lib.cpmh_hull_cycle.argtypes = i32, [fp, i32, fp]
        lib.cpmh_smallest_rectangle.restype, lib.cpmh_smallest_rectangle.argtypes = None, [fp, i32, fp]
        lib.cpmh_tf_difference.restype = i32
        lib.cpmh_tf_difference.argtypes = [dp, fp, i32, dp, fp, i32, C.c_float, i32, fp, fp]
        _hostmath_lib = lib
    return _hostmath_lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def convex_hull_2d(points):
    """The vertex cycle the rectangle search walks (cpm_host::hullCycle, rule H; ref lightcl/convexhull2d.cpp:38-130)."""
    pts = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 2))
    out = np.zeros((len(pts) + 2, 2), np.float32)
    n = _hostmath().cpmh_hull_cycle(_fp(pts), len(pts), _fp(out))
    return [(out[i, 0], out[i, 1]) for i in range(n)]


def minimum_bounding_rectangle(hull):
    """(corner, side0, side1) of the smallest rectangle along an edge of the cycle (ref lightcl/orientedboundingbox2d.cpp:40-78)."""
    pts = np.ascontiguousarray(np.asarray(hull, np.float32).reshape(-1, 2))
    out = np.zeros(6, np.float32)
    _hostmath().cpmh_smallest_rectangle(_fp(pts), len(pts), _fp(out))
    return out[0:2].copy(), out[2:4].copy(), out[4:6].copy()


def fit_plane_aligned_obb(points, plane_point, plane_normal):
    """(origin, u, v) of the minimum-area rectangle, in the light plane, covering the projected points
    (cpm_host::fitLightRectangle; ref lightcl/orientedboundingbox2d.cpp:80-100)."""
    pts = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 3))
    through = np.ascontiguousarray(np.asarray(plane_point, np.float32))
    n = np.ascontiguousarray(_normalize(plane_normal))
    out = np.zeros(9, np.float32)
    _hostmath().cpmh_fit_light_rectangle(_fp(pts), len(pts), _fp(through), _fp(n), _fp(out))
    return out[0:3].copy(), out[3:6].copy(), out[6:9].copy()


def tf_difference_points(tf_points, prev_points, eps=1e-4, associated=False):
    """Break-point list of |TF_new - TF_old|, zero-padded at 0 and 1 (cpm_host::tfDifference, rule D; ref importancesamplingcl/
    processors/minmaxuniformgrid3dimportanceclprocessor.cpp:364-501).  tf_points / prev_points: lists of (position, r, g, b, a).
    Returns (positions, colors4)."""
    def split(points):
        pts = sorted(points, key=lambda q: float(q[0]))
        pos = np.ascontiguousarray([float(q[0]) for q in pts], np.float64)
        rgba = np.ascontiguousarray([[np.float32(c) for c in q[1:5]] for q in pts], np.float32).reshape(-1, 4)
        return pos, rgba
    pa, ca = split(tf_points)
    pb, cb = split(prev_points)
    cap = len(pa) + len(pb) + 2
    out_pos, out_rgba = np.zeros(cap, np.float32), np.zeros((cap, 4), np.float32)
    dp = C.POINTER(C.c_double)
    n = _hostmath().cpmh_tf_difference(pa.ctypes.data_as(dp), _fp(ca), len(pa), pb.ctypes.data_as(dp), _fp(cb), len(pb),
                                       float(eps), int(bool(associated)), _fp(out_pos), _fp(out_rgba))
    if n < 0:
        raise ValueError("tf_difference_points: one of the two transfer functions is empty")
    return out_pos[:n].copy(), out_rgba[:n].copy()


# --------------------------------------------------------------------------- frame

class PhotonFrame:
    """One light, one volume: owns the device buffers of the path and runs its stages.

    n_side: an int (n_side x n_side emission lattice) or a pair (nx, ny).
    photon_range = (lo, hi) restricts this instance to a contiguous shard of the N = nx * ny
    photons (multi-GPU: one shard per rank); photon_indices (ascending global indices, e.g.
    sharding.shard_tiles: the lattice's 4096-sample tiles dealt round-robin) to any other shard.
    Photon i keeps light sample i and RNG stream i of the unsharded run, so results do not depend
    on the number or shape of the shards; local photon j is global photon photon_indices[j].
    """

    def __init__(self, ctx: B.Context, volume, tf_rgba, n_side: int, grid_dims, *,
                 light_travel_direction=(0.0, 0.0, 1.0), light_distance: float = 2.0,
                 radiance=(1.0, 1.0, 1.0), radius_voxels: float = 1.0, max_interactions: int = 1,
                 channels: int = 1, photon_range=None, point_light_position=None, seed: int = 0,
                 shading_type: int = B.CPM_PHASE_HENYEY_GREENSTEIN, material=(0.0, 0.0, 0.0, 0.0), light_plane=None,
                 mesh_intersection=None, emit_in_tracer: bool = False, photon_indices=None, records_layout=None):
        torch = ctx.torch
        self.ctx = ctx
        self.records_layout = records_layout
        self.torch = torch
        dev = ctx.device
        self._vol_is_own = not isinstance(volume, B.Volume)   # created here (set_volume may recycle it) or the caller's
        self.vol = volume if isinstance(volume, B.Volume) else ctx.volume_create(volume)
        self.tf = tf_rgba if isinstance(tf_rgba, B.TransferFunction) else ctx.tf_create(tf_rgba)
        nx, ny = (n_side, n_side) if isinstance(n_side, int) else n_side
        self.n_total = nx * ny
        if photon_indices is not None:
            if photon_range is not None:
                raise ValueError("photon_range and photon_indices are alternatives")
            gidx = np.ascontiguousarray(photon_indices, dtype=np.int64)
            if gidx.size and (gidx.min() < 0 or gidx.max() >= self.n_total or np.any(np.diff(gidx) <= 0)):
                raise ValueError("photon_indices must be ascending indices into the nx * ny lattice")
            contiguous = gidx.size > 0 and int(gidx[-1]) - int(gidx[0]) + 1 == gidx.size
            if contiguous:  # a range after all (one rank, or a shard of one tile)
                photon_range, photon_indices = (int(gidx[0]), int(gidx[-1]) + 1), None
            elif gidx.size == 0:
                photon_range, photon_indices = (0, 0), None
        lo, hi = photon_range if photon_range is not None else (0, self.n_total)
        self.lo, self.hi = lo, hi
        self.n = hi - lo
        self.global_index = None   # device int64 [n]: global index of local photon j (None = lo + j)
        if photon_indices is not None:
            self.global_index = torch.from_numpy(gidx).to(dev)
            self.n = int(gidx.size)
            emit_in_tracer = False   # the in-register emitter addresses lattice sample first_sample + thread
        self.I = max_interactions
        self.aabb = S.UNIT_CUBE_AABB
        vdims = self.vol.dims
        self.grid = B.default_grid_desc(grid_dims, channels)
        self.cells = int(grid_dims[0]) * int(grid_dims[1]) * int(grid_dims[2])
        self.radius = S.photon_radius_texture(vdims, radius_voxels)
        self.scale = B.relative_irradiance_scale(self.radius, float(self.n_total))

        # E1: emission lattice, E2: light plane, E3/E5: light samples, E4: entry/exit
        samples = ctx.uniform_samples_2d(nx, ny)
        samples = (samples[lo:hi] if self.global_index is None else samples.index_select(0, self.global_index)).contiguous()
        # This driver owns its light, so the tracer can evaluate the emission chain itself (emit_in_tracer=True ->
        # cpm_trace_emitted: same device functions, same bits, 40 input bytes per photon not read and no buffers needed by
        # the trace).  Off by default: measured at config 2 the tracer is bound by VALU issue, not by its input bytes
        # (34.5 us emitted against 33.9 us from the buffers -- the lattice fmod, the pdf divisions and the box test cost what
        # the loads saved).  The buffers stay what every other consumer reads (the importance pass of the correlated update).
        self.emitter = None
        if point_light_position is not None:
            self.light_samples = ctx.point_light_samples(samples, radiance, point_light_position)
            if emit_in_tracer and mesh_intersection is None:
                self.emitter = B.point_emitter(nx, ny, radiance, point_light_position, first_sample=lo)
        else:
            d = np.asarray(light_travel_direction, np.float32) if light_plane is not None else _normalize(light_travel_direction)
            if light_plane is not None:  # (origin, u, v, area) fitted elsewhere (e.g. by the C++ host layer)
                o, u, v, area = light_plane
            else:
                origin = np.array([0.5, 0.5, 0.5], np.float32) - np.float32(light_distance) * d
                o, u, v = fit_plane_aligned_obb(S.UNIT_CUBE_VERTICES, origin, d)
                area = float(np.float32(np.linalg.norm(u)) * np.float32(np.linalg.norm(v)))
            self.light_plane = (o, u, v, area, d)
            self.light_samples = ctx.directional_light_samples(samples, radiance, d, o, u, v, area)
            if emit_in_tracer and mesh_intersection is None:
                self.emitter = B.directional_emitter(nx, ny, radiance, d, o, u, v, area, first_sample=lo)
        if mesh_intersection is not None:  # (vertices [n,3] f32, indices int32): the proxy-mesh variant of E4
            vtx = torch.from_numpy(np.ascontiguousarray(mesh_intersection[0], np.float32)).to(dev)
            idx = torch.from_numpy(np.ascontiguousarray(mesh_intersection[1], np.int32)).to(dev)
            self.isect = ctx.light_sample_mesh_intersection(vtx, idx, self.light_samples)
        else:
            self.isect = ctx.light_sample_box_intersection(self.light_samples, self.aabb)

        # R2: per-photon MWC64X streams (glibc srand(seed) bases, gap 2^40), seeded over the
        # unsharded index range so that stream i is the same whatever the shard
        bases = B.glibc_rand_sequence(seed, self.n_total)
        st = np.zeros((self.n_total, 2), dtype=np.uint32)
        st[:, 0] = bases
        full = torch.from_numpy(st.view(np.int32)).to(dev).view(torch.int32)
        ctx.seed_streams(full)
        self.rng = (full[lo:hi] if self.global_index is None else full.index_select(0, self.global_index)).contiguous()
        self.rng_initial = self.rng.clone()

        self.params = B.TraceParams()
        for i in range(4):
            self.params.material[i] = material[i]
        self.params.step_size = 1.0 / max(vdims)
        self.params.photon_offset = 0
        self.params.n_light_samples = self.n
        self.params.max_interactions = self.I
        self.params.total_photons = self.n
        self.params.shading_type = shading_type
        self.params.flags = 0

        f32 = torch.float32
        self.photons = torch.empty((self.n * self.I, 8), dtype=f32, device=dev)
        self.order = torch.empty(self.n * self.I, dtype=torch.int32, device=dev)
        self.cell_start = torch.empty(self.cells + 1, dtype=torch.int32, device=dev)
        self.sorted = torch.empty((self.n * self.I, 4 if channels == 1 else 8), dtype=f32, device=dev)
        self.light_volume = torch.zeros((self.cells, channels) if channels > 1 else (self.cells,), dtype=f32, device=dev)
        self.adaptive_order = True   # full traces take their chunks in the order of their measured costs (cpm_trace_order_*)
        self.trace_order = None
        self._traces_since_order = 0
        self._order_stale = False
        self.brick_table = None   # cpm_bin_fast's table and records (every photon in all the bricks it reaches), allocated on first use
        self.sorted_fast = None
        self.photon_layout = ctx.photon_layout()   # how self.photons is laid out: the context's layout (cpm_set_photon_layout), or set_planar_records
        self._records8 = None
        # records_layout="planar": THIS frame's record buffers are two planes whatever the context's default -- described to the library
        # buffer by buffer (cpm_records_describe), not through a context-wide mode
        self._described = []
        if records_layout == "planar":
            self.photon_layout = B.CPM_PHOTONS_PLANAR
            self._describe(self.photons)
        elif records_layout not in (None, "context"):
            raise ValueError("records_layout: None (the context's) or 'planar'")

    def _describe(self, buf, replaces=None):
        """buf holds this frame's N * I records in the frame's layout: say so to the library when the layout is this frame's own
        (replaces: the buffer it takes the place of, whose description goes)."""
        if getattr(self, "records_layout", None) == "planar":
            if replaces is not None:
                self.ctx.records_forget(replaces)
                self._described = [b for b in self._described if b is not replaces]
            self.ctx.records_describe(buf, B.CPM_PHOTONS_PLANAR, self.n * self.I)
            self._described.append(buf)
        return buf

    def forget_described(self):
        for b in self._described:
            self.ctx.records_forget(b)
        self._described = []

    def __del__(self):
        try:   # (a described buffer's entry must not outlive the buffer: its memory may come back as something else)
            if getattr(self, "_described", None) and self.ctx.h:
                self.forget_described()
        except Exception:
            pass

    def set_planar_records(self, on=True):
        """The tracer writes -- and the brick bin reads -- the two-plane record layout (CPM_TRACE_PHOTONS_PLANAR / cpm_bin_fast_layout):
        the same records, position + first power channel in one plane; self.photons then holds that layout and `records()` gives
        the float8 form.  For frames of trace -> bin_fast -> gather_fast; everything else on this class reads float8 records."""
        self.photon_layout = B.CPM_PHOTONS_PLANAR if on else B.CPM_PHOTONS_INTERLEAVED
        if on:
            self.params.flags |= B.CPM_TRACE_PHOTONS_PLANAR
        else:
            self.params.flags &= ~B.CPM_TRACE_PHOTONS_PLANAR

    def records(self):
        """The photon records as float8 rows (what the `photons` port carries), converted when self.photons is planar."""
        if self.photon_layout == B.CPM_PHOTONS_INTERLEAVED:
            return self.photons
        if self._records8 is None:
            self._records8 = self.torch.empty_like(self.photons)
        self.ctx.photons_convert(self.photons, B.CPM_PHOTONS_PLANAR, self._records8, B.CPM_PHOTONS_INTERLEAVED, self.n * self.I)
        return self._records8

    # stages
    #: every so many full traces one is measured (what each chunk cost) and the order re-sorted from it (cpm_trace_order_update);
    #: so is the first one, and the first after a change of what is traced (invalidate_trace_order)
    TRACE_ORDER_EVERY = 256
    TRACE_ORDER_AT_LEAST_APART = 32

    def trace(self, recompute_indices=None, n_recompute=0):
        full = recompute_indices is None and self.adaptive_order and self.n > 0   # (an empty shard traces nothing and needs no order)
        measure = full and (self._traces_since_order == 0 or self._traces_since_order >= self.TRACE_ORDER_EVERY or
                            (self._order_stale and self._traces_since_order >= self.TRACE_ORDER_AT_LEAST_APART))
        if full:
            if self.trace_order is None:
                self.trace_order = self.ctx.trace_order_create(self.n)
            self.ctx.trace_set_order(self.trace_order, measure)
        try:
            if self.emitter is not None:
                self.ctx.trace_emitted(self.vol, self.tf, self.aabb, self.params, self.emitter, self.rng, self.photons,
                                       recompute_indices=recompute_indices, n_recompute=n_recompute)
            else:
                self.ctx.trace(self.vol, self.tf, self.aabb, self.params, self.light_samples, self.isect, self.rng, self.photons,
                               recompute_indices=recompute_indices, n_recompute=n_recompute)
        finally:
            if full:
                self.ctx.trace_set_order(None)
        if full:
            if measure:
                self.trace_order.update()   # two small launches behind the measured trace
                self._traces_since_order = 0
                self._order_stale = False
            self._traces_since_order += 1

    def invalidate_trace_order(self):
        """Volume / transfer function / light changed: a full trace soon is measured and followed by a re-sort of its chunks (a
        stale order is still a valid order; measuring costs what ten launches gain, so not more often than every
        TRACE_ORDER_AT_LEAST_APART-th launch)."""
        self._order_stale = True

    def bin(self):
        self.ctx.bin(self.photons, self.n * self.I, self.grid, self.order, self.cell_start, self.sorted)

    def gather(self, accumulate=False, out=None):
        self.ctx.gather(self.sorted, self.cell_start, self.n * self.I, self.grid, self.radius, self.scale,
                        self.light_volume if out is None else out, accumulate=accumulate)

    # tolerance-mode formulation: brick bin + LDS-tile gather (cpm_bin_fast / cpm_gather_fast)
    def bin_fast(self):
        if self.brick_table is None:
            entries = self.ctx.fast_table_entries(self.grid, self.n * self.I)
            self.brick_table = self.torch.zeros(entries, dtype=self.torch.int32, device=self.ctx.device)
        cap = max(self.ctx.fast_record_capacity(self.grid, self.n * self.I, self.radius), 1)
        if self.sorted_fast is None or self.sorted_fast.shape[0] < cap:   # (a progressive radius schedule only ever shrinks it)
            self.sorted_fast = self.torch.empty((cap, 4 if self.grid.channels == 1 else 8), dtype=self.torch.float32, device=self.ctx.device)
        self.ctx.bin_fast(self.photons, self.n * self.I, self.grid, self.radius, self.brick_table, self.sorted_fast, layout=self.photon_layout)

    def gather_fast(self, accumulate=False, out=None, nonzero_bricks=None):
        self.ctx.gather_fast(self.sorted_fast, self.brick_table, self.n * self.I, self.grid, self.radius, self.scale,
                             self.light_volume if out is None else out, accumulate=accumulate, nonzero_bricks=nonzero_bricks)

    def gather_fast_segment(self, segment):
        """The gather of a shard that is not the display GPU: its non-zero 4x4x4 bricks straight into the brick-list segment of an opened
        ticket (binding.BricklistReduce.open) -- no dense light volume, no zeros (cpm_gather_fast_segment)."""
        self.ctx.gather_fast_segment(self.sorted_fast, self.brick_table, self.n * self.I, self.grid, self.radius, self.scale, segment)

    def frame_fast(self):
        """The hot path in tolerance mode: trace -> brick bin -> tile gather."""
        self.trace()
        self.bin_fast()
        self.gather_fast()
        return self.light_volume

    def splat(self, out=None, all_interactions=False):
        """Reference formulation (atomic splat), for comparison: clear + splat.
        The reference's full splat adds interaction 0 only (its guard compares against N although the launch covers N x I
        work-items: ref cl/photonstolightvolume.cl:154-158, processor/photontolightvolumeprocessorcl.cpp:304,380; SURVEY
        Q1) -- the default here; bin + gather cover all N x I records (what the reference's own incremental splat does,
        :192-201).  all_interactions=True splats all records, the counterpart of the gather at I > 1."""
        out = self.light_volume if out is None else out
        out.zero_()
        self.ctx.splat(self.photons, self.n * self.I if all_interactions else self.n, self.grid, self.radius, self.scale, out)
        return out

    def frame(self):
        """The hot path: trace -> bin -> gather."""
        self.trace()
        self.bin()
        self.gather()
        return self.light_volume

    # The library allocates nothing after its first call, so the whole frame (~14 launches) is
    # capturable into a HIP graph.  Measured on MI355X / ROCm 7.2 the replay is SLOWER than eager
    # launches for this frame (0.299 vs 0.251 ms): kept as an option, not the default.
    def capture(self):
        torch = self.torch
        self.frame()                      # warm-up: scratch arenas reach their final size
        torch.cuda.synchronize(self.ctx.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.trace()
            self.bin()
            self.gather()
        self._graph = g
        return g

    def replay(self):
        self._graph.replay()
        return self.light_volume


# --------------------------------------------------------------------------- progressive refinement

def progressive_sphere_radius(radius: float, iteration: int, alpha: float) -> float:
    """r_(i+1) = r_i ((i + alpha) / (i + 1))^(1/3), in double (Knaus & Zwicker eq. 20;
    ref progressivephotonmapping/photondata.cpp:72-77)."""
    return radius * math.pow((float(iteration) + alpha) / (1.0 + float(iteration)), 1.0 / 3.0)


class ProgressivePhotonMapper(PhotonFrame):
    """Progressive photon mapping on the path: every iteration traces the SAME light samples with the RNG streams where the
    previous iteration left them (-D PROGRESSIVE_PHOTON_MAPPING: the state is written back, ref cl/photontracer.cl:211-215),
    with the radius shrunk by the schedule above (PhotonData::advanceToNextIteration, ref photondata.cpp:67-79, driven by
    ProgressivePhotonTracerCL::process, ref processor/progressivephotontracercl.cpp:252-260), and the light volume becomes
    the running average of the iterations' estimates: L_i = L_(i-1) + (E_i - L_(i-1)) / i  (cpm_mix_buffers: the OpenCL mix).
    The reference re-splats each iteration's photons and leaves the averaging to the consumer; the average is this build's."""

    def __init__(self, *args, alpha: float = 0.5, formulation: str = "fast", **kw):
        super().__init__(*args, **kw)
        self.alpha = alpha
        self.formulation = formulation
        self.iteration = 0
        self.radius0 = self.radius
        self.estimate = self.torch.zeros_like(self.light_volume)
        self.params.flags |= B.CPM_TRACE_PROGRESSIVE

    def reset(self):
        """Light / camera / TF / volume changed (ref tracercl.cpp:240-247): iteration 0, original streams and radius."""
        self.iteration = 0
        self.radius = self.radius0
        self.rng.copy_(self.rng_initial)

    def iterate(self):
        if self.iteration == 0:
            self.radius = self.radius0
            self.iteration = 1
        else:
            self.radius = progressive_sphere_radius(self.radius, self.iteration, self.alpha)
            self.iteration += 1
        self.scale = B.relative_irradiance_scale(self.radius, float(self.n_total))
        self.params.iteration = self.iteration
        self.trace()
        out = self.light_volume if self.iteration == 1 else self.estimate
        fast = self.formulation == "fast" and self.ctx.gather_fast_supported(self.grid, self.radius)
        if fast:
            self.bin_fast()
            self.gather_fast(out=out)
        else:
            self.bin()
            self.gather(out=out)
        if self.iteration > 1:
            self.ctx.mix_buffers(self.light_volume, self.estimate, 1.0 / self.iteration, self.light_volume)
        return self.light_volume


# --------------------------------------------------------------------------- correlated re-trace (C7 + G5)

class CorrelatedPhotonMapper(PhotonFrame):
    """PhotonFrame plus the correlated update after a transfer-function edit or a time step:
    min/max bricks -> TF-difference importance per brick -> per-photon importance by DDA ->
    select (threshold + count + iota + sort by importance) -> index-ordered re-trace of the n
    most important photons with their ORIGINAL RNG streams -> light-volume update
    (ProgressivePhotonTracerCL::process importance branch, ref
    progressivephotonmapping/processor/progressivephotontracercl.cpp:265-541, and
    PhotonToLightVolumeProcessorCL::process, ref .../photontolightvolumeprocessorcl.cpp:196-354).
    """

    def __init__(self, *args, region: int = 8, max_incremental_percent: float = 100.0,
                 incremental_threshold_percent: float = 50.0, fix_exit_point: bool = False, tf_points=None,
                 exact_update: bool = False, formulation: str = "fast", **kw):
        super().__init__(*args, **kw)
        # full light volumes: "fast" = brick bin + one-launch gather (tolerance mode; the reference's own update is two atomic
        # splats on top of it, within the same tolerance), "exact" = cell sort + sequential gather -- forced by exact_update,
        # whose touched-brick re-gather must land on a full gather's bits
        self.formulation = "exact" if exact_update else formulation
        torch, dev, ctx = self.torch, self.ctx.device, self.ctx
        self.region = region
        self.max_incremental_percent = max_incremental_percent
        self.incremental_threshold_percent = incremental_threshold_percent
        self.fix_exit_point = fix_exit_point
        # exact_update: re-bin and re-gather exactly the bricks a changed photon touches (bit-identical to a full frame)
        # instead of the reference's -old / +new atomic splats (cheaper, within fp32 tolerance, not reproducible)
        self.exact_update = exact_update
        self.brick_mask = None
        # multi-GPU delta path: when set (a zero-initialised uint8 tensor, one entry per 4x4x4-voxel brick), every light-volume
        # update marks the bricks it changed -- what cpm_allreduce_grid_bricks sums over the ranks
        self.touched_mask = None
        self.tf_points = list(tf_points) if tf_points is not None else list(S.WORKSPACE_TF_POINTS)
        vd = self.vol.dims
        self.brick_dims = tuple((d + region - 1) // region for d in vd)
        nb = self.brick_dims[0] * self.brick_dims[1] * self.brick_dims[2]
        self.minmax = torch.zeros((nb, 2), dtype=torch.int16, device=dev)
        ctx.volume_minmax(self.vol, region, self.minmax)          # VolumeMinMaxCLProcessor
        self.importance_grid = torch.zeros(nb, dtype=torch.float32, device=dev)
        self.importance = torch.empty(self.n, dtype=torch.int32, device=dev)
        ctx.reset_importance(self.importance, 0, self.n)
        self.indices = torch.empty(self.n, dtype=torch.int32, device=dev)
        self.n_changed = torch.zeros(1, dtype=torch.int32, device=dev)
        self.prev_photons = None
        self.remaining_offset = 0
        self.remaining = -1
        self.n_recomputed = -1
        self.last_path = None
        # the update without a host round trip (cpm_selection_*, cpm_trace_selected, cpm_splat_delta): taken when every changed
        # photon is traced in the same evaluation (budget = 100 %, the default) and the update is the reference's add-remove
        self.selection = None
        self.old_photons = None
        self.fused = True
        self.retrace_in_importance_pass = True   # cpm_photon_importance_retrace (False: select, compact, then cpm_trace_selected)

    def full_frame(self):
        """Light / everything changed: full trace, bin + gather, snapshot (tracercl.cpp:541-560)."""
        self.trace()
        self._full_light_volume()
        self.ctx.reset_importance(self.importance, 0, self.n)
        # the previous-photon snapshot (a whole-buffer copy per evaluation in the reference, processorcl.cpp:343-352) is only
        # kept where the legacy add-remove reads it; the fused update's tracer keeps the records it replaces itself
        self.have_frame = True
        if self._fused_configured():
            self._prev_stale = True
        else:
            self.prev_photons = self._describe(self.photons.clone(), replaces=self.prev_photons)
            self._prev_stale = False
        self.n_recomputed = -1
        self.remaining, self.remaining_offset = 0, 0
        self.last_path = "full"
        return self.light_volume

    def _full_light_volume(self):
        if self.formulation == "fast" and self.ctx.gather_fast_supported(self.grid, self.radius):
            self.bin_fast()
            self.gather_fast()
        else:
            self.bin()
            self.gather()

    def _occupancy(self):
        """The importance grid's occupancy bits, written by the launch that writes the grid and handed to the selection."""
        if getattr(self, "_occupancy_bits", None) is None:
            nb = self.importance_grid.numel()
            self._occupancy_bits = self.torch.zeros(2 * ((nb + 63) // 64), dtype=self.torch.int32, device=self.ctx.device)
        return self._occupancy_bits

    def set_transfer_function(self, tf_points, width=1024, moved=None):
        """A TF edit: updates the LUT and the importance grid (MinMaxUniformGrid3DImportanceCLProcessor)."""
        self.invalidate_trace_order()
        pos, col = tf_difference_points(tf_points, self.tf_points)
        self.tf_points = list(tf_points)
        self.tf.update(S.tf_from_points(tf_points, width))
        nb = self.importance_grid.numel()
        self.ctx.importance_tf(self.minmax, nb, pos, col, self.importance_grid, occupancy=self._occupancy())
        return pos, col

    def set_volume(self, voxels):
        """A time step: new voxel data (same shape).  Computes, on the GPU, the per-brick mean |v_new - v_old|
        (CPU in the reference: DynamicVolumeDifferenceAnalysis), the new min/max bricks and the time-varying
        importance = difference x TF importance over the union of the old and new brick ranges
        (classifyTimeVaryingMinMaxUniformGrid3DImportanceKernel), then swaps the volume in.

        `voxels`: an array / device tensor (copied into a volume this mapper owns), or a binding.Volume created
        beforehand, which is adopted as it is -- the element of a sequence whose device representation already exists
        (Inviwo caches a VolumeCL per sequence element the same way), no copy and no re-layout in the step."""
        ctx, torch = self.ctx, self.torch
        self.invalidate_trace_order()
        nb = self.importance_grid.numel()
        if getattr(self, "_minmax_next", None) is None:
            self._minmax_next = torch.zeros_like(self.minmax)
            self._diff = torch.zeros(nb, dtype=torch.float32, device=ctx.device)
            self._vol_next = None            # a volume of this mapper's own to copy raw voxels into
        adopted = isinstance(voxels, B.Volume)
        if adopted:
            if tuple(voxels.dims) != tuple(self.vol.dims) or int(voxels.desc.dtype) != int(self.vol.desc.dtype):
                raise ValueError("set_volume: the time step differs from the current volume in shape or type")
            nxt = voxels
        elif self._vol_next is None:
            nxt = ctx.volume_create(voxels)
        else:
            nxt = self._vol_next
            nxt.update(voxels)
        self._vol_next = nxt
        ctx.volume_step(self.vol, self._vol_next, self.region, self._diff, self._minmax_next)   # difference + min/max, one pass
        # TF unchanged: importance of a range = the TF itself (updateTransferFunctionData), zero-padded to [0, 1]
        pts = sorted(self.tf_points)
        pos = [p[0] for p in pts]
        col = [list(p[1:]) for p in pts]
        if pos[0] > 0.0:
            pos.insert(0, 0.0); col.insert(0, col[0])
        if pos[-1] < 1.0:
            pos.append(1.0); col.append(col[-1])
        ctx.importance_tf(self._minmax_next, nb, np.asarray(pos, np.float32), np.asarray(col, np.float32), self.importance_grid,
                          prev_minmax=self.minmax, volume_diff=self._diff, occupancy=self._occupancy())
        # swap in; the volume swapped out is reused for the next raw-voxel step only if this mapper created it
        previous, previous_is_own = self.vol, self._vol_is_own
        self.vol, self._vol_is_own = nxt, not adopted
        self._vol_next = previous if previous_is_own else None
        self.minmax, self._minmax_next = self._minmax_next, self.minmax

    def _fused_configured(self):
        return self.fused and not self.exact_update and self.max_incremental_percent >= 100.0 and self.emitter is None

    def _fused_available(self):
        return self._fused_configured() and getattr(self, "have_frame", False)

    def correlated_update_fused(self):
        """The same evaluation with the count kept on the device: importance + threshold + tile lists in one launch, the lists
        lined up by a second, the tracer and the - old / + new splat launched over the budget and bounded by the device count.
        The host reads the count once, after everything is enqueued (from the selection's mailbox).  Returns n re-traced."""
        ctx, torch = self.ctx, self.torch
        n_total = self.n
        if self.selection is None:
            self.selection = ctx.selection_create(n_total)
            self.old_photons = torch.empty((self.I * n_total, 8), dtype=torch.float32, device=ctx.device)
        sel = self.selection
        sel.begin()
        if getattr(self, "_occupancy_bits", None) is not None:   # (the bits of the grid's last importance_tf launch)
            sel.set_occupancy(self.importance_grid, self._occupancy_bits)
        self.params.flags = 0                            # correlated: RNG state is NOT written back
        if self.retrace_in_importance_pass:
            # detector + threshold + tracer in one launch; the replaced records stay at the photons' own indices
            sel.photon_importance_retrace(self.importance_grid, self.brick_dims, (float(self.region),) * 3, list(self.vol.desc.texture_to_index),
                                          self.vol, self.tf, self.aabb, self.params, self.light_samples, self.isect, self.importance,
                                          self.rng, self.photons, self.old_photons, fix_exit_point=self.fix_exit_point)
            sel.finish(self.indices)
            old_stride = 0
        else:
            sel.photon_importance(self.importance_grid, self.brick_dims, (float(self.region),) * 3, list(self.vol.desc.texture_to_index),
                                  self.photons, 0, self.light_samples, self.isect, n_total, self.I, n_total, self.importance,
                                  fix_exit_point=self.fix_exit_point)
            sel.finish(self.indices)
            ctx.trace_selected(self.vol, self.tf, self.aabb, self.params, self.light_samples, self.isect, self.indices, sel, n_total,
                               self.rng, self.photons, old_photons=self.old_photons, reset_importances=self.importance)
            old_stride = n_total
        max_recomp = int(self.n * (self.incremental_threshold_percent / 100.0))
        if self.touched_mask is not None:
            self.touched_mask.zero_()
        ctx.splat_delta(self.old_photons, old_stride, self.photons, self.indices, sel, n_total, self.grid, self.radius, self.scale,
                        self.n, self.I, self.light_volume, apply_below=max(max_recomp, 1), brick_mask=self.touched_mask)
        n = sel.count()                                  # the one host read, behind everything enqueued
        self.n_recomputed = n
        self.remaining, self.remaining_offset = 0, n
        if n == 0:
            self.last_path = "unchanged"
        elif n < max_recomp:
            self.last_path = "incremental"
        else:                                            # the delta launch stood aside: rebuild (processorcl.cpp:299-339)
            self._full_light_volume()
            self.last_path = "full"
            if self.touched_mask is not None:
                self.touched_mask.fill_(1)
        # (no snapshot to refresh: the tracer kept the records it replaced; prev_photons stays allocated for the legacy path only)
        self._prev_stale = True
        return n

    def correlated_update(self):
        """One evaluation of the importance branch + the light-volume processor.  Returns n re-traced."""
        ctx, torch = self.ctx, self.torch
        n_total = self.n
        vd = self.vol.dims
        if self._fused_available():
            return self.correlated_update_fused()
        if getattr(self, "_prev_stale", False) and getattr(self, "have_frame", False):   # the fused path does not maintain the snapshot
            self.prev_photons = self._describe(self.photons.clone(), replaces=self.prev_photons)
            self._prev_stale = False
        ctx.photon_importance(self.importance_grid, self.brick_dims, (float(self.region),) * 3,
                              list(self.vol.desc.texture_to_index), self.photons, 0, self.light_samples, self.isect,
                              n_total, self.I, n_total, self.importance, fix_exit_point=self.fix_exit_point)
        # The changed photons, ascending: one radix pass over a flag; the count is read once (the reference's single
        # host sync, tracercl.cpp:374).  Ranking them by importance only matters when they do not all fit this
        # evaluation's budget -- otherwise the batch is re-sorted by index for the trace anyway (:467-473).
        ctx.select_changed(self.importance, self.indices, self.n_changed)
        n_changed = int(self.n_changed.item())
        self.remaining_offset = 0
        if self.remaining < 0 or n_changed > 0:
            self.remaining = n_changed
        max_update = int((self.max_incremental_percent / 100.0) * n_total)
        if n_changed <= max_update:
            return self._retrace_batch(ranked=False)
        ctx.select_recompute(self.importance, self.indices, self.n_changed)   # thresholdKernel ... sortIndicesByImportance
        return self._retrace_batch()

    def continue_update(self):
        """Progressive continuation on the 100 ms timer (tracercl.cpp:387-419,534-540)."""
        if self.remaining <= 0:
            return 0
        return self._retrace_batch()

    def _retrace_batch(self, ranked=True):
        """ranked: self.indices / self.importance are sorted by importance (cpm_select_recompute); otherwise
        self.indices[:remaining] is the complete, ascending list of changed photons (cpm_select_changed)."""
        ctx, torch = self.ctx, self.torch
        n_total = self.n
        max_update = int((self.max_incremental_percent / 100.0) * n_total)
        n = min(self.remaining, max_update)
        self.n_recomputed = n
        if n > 0:
            idx = self.indices[self.remaining_offset:self.remaining_offset + n].contiguous()
            if ranked:
                ctx.sort_keys(idx, 0)                        # ascending index = emission-lattice order (:467-473)
            self.params.flags = 0                            # correlated: RNG state is NOT written back
            self.trace(recompute_indices=idx, n_recompute=n)
            if ranked:  # importance is sorted alongside the indices: entries [offset, offset+n) belong to the re-traced photons
                ctx.reset_importance(self.importance, self.remaining_offset, n)
            else:       # every changed photon was re-traced
                ctx.reset_importance(self.importance, 0, n_total)
            self._update_light_volume(idx, n)
        self.remaining_offset += n
        self.remaining -= n
        return n

    def _update_light_volume(self, idx, n):
        ctx = self.ctx
        max_recomp = int(self.n * (self.incremental_threshold_percent / 100.0))
        if self.prev_photons is not None and 0 < n < max_recomp and self.exact_update:
            gd = self.grid.dims
            nb = ((gd[0] + 3) // 4) * ((gd[1] + 3) // 4) * ((gd[2] + 3) // 4)
            if self.brick_mask is None:
                self.brick_mask = self.torch.empty(nb, dtype=self.torch.uint8, device=ctx.device)
            self.brick_mask.zero_()
            ctx.mark_touched_bricks(self.prev_photons, idx, n, self.n, self.I, self.grid, self.radius, self.brick_mask)
            ctx.mark_touched_bricks(self.photons, idx, n, self.n, self.I, self.grid, self.radius, self.brick_mask)
            self.bin()
            ctx.gather_bricks(self.sorted, self.cell_start, self.n * self.I, self.grid, self.radius, self.scale, self.brick_mask,
                              self.light_volume)
            if self.touched_mask is not None:   # multi-GPU delta reduce: the re-gathered bricks are the ones that changed
                self.touched_mask.copy_(self.brick_mask)
            self.last_path = "exact incremental"
        elif self.prev_photons is not None and 0 < n < max_recomp:
            # incremental: remove the old contributions, add the new ones (processorcl.cpp:196-298)
            if self.touched_mask is not None:
                self.touched_mask.zero_()
                ctx.mark_touched_bricks(self.prev_photons, idx, n, self.n, self.I, self.grid, self.radius, self.touched_mask)
                ctx.mark_touched_bricks(self.photons, idx, n, self.n, self.I, self.grid, self.radius, self.touched_mask)
            ctx.splat_selected(self.prev_photons, idx, n, self.grid, self.radius, self.scale, -1.0, self.n, self.I, self.light_volume)
            ctx.splat_selected(self.photons, idx, n, self.grid, self.radius, self.scale, 1.0, self.n, self.I, self.light_volume)
            self.last_path = "incremental"
        else:
            self._full_light_volume()
            self.last_path = "full"
            if self.touched_mask is not None:
                self.touched_mask.fill_(1)
        # snapshot for the next add-remove (a whole-buffer copy in the reference, :343-352): after a partial re-trace
        # only the re-traced photons differ from the snapshot, so only they move
        if self.last_path == "full" or self.prev_photons is None:
            self.prev_photons = self._describe(self.photons.clone(), replaces=self.prev_photons)
        else:
            ctx.snapshot_selected_photons(self.photons, idx, n, self.n, self.I, self.prev_photons)
