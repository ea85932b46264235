"""ctypes binding of libcpm_hip.so (include/cpm/cpm.h).

PyTorch is used for device memory and streams only: every buffer handed to the
library is a ``torch.Tensor`` on the context's device, passed as its raw device
pointer; work is enqueued on torch's current stream.  There is no CPU
fallback: when the shared library is missing the import fails loudly, and
``Context()`` raises when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "libcpm_hip.so"

CPM_OK = 0
CPM_U8, CPM_U16, CPM_F32 = 0, 1, 2
CPM_TRACE_PROGRESSIVE = 1
CPM_TRACE_NO_SINGLE_SCATTERING = 2
CPM_TRACE_PHOTONS_PLANAR = 4
CPM_PHOTONS_INTERLEAVED, CPM_PHOTONS_PLANAR = 0, 1
CPM_PHASE_HENYEY_GREENSTEIN, CPM_PHASE_ISOTROPIC = 0, 1

#: every symbol include/cpm/cpm.h declares -- the core: one entry point per call site of the reference's host code (checked by tests/test_abi.py
#: against the header)
CORE_SYMBOLS = [
    "cpm_create", "cpm_destroy", "cpm_last_error_string", "cpm_abi_version", "cpm_glibc_rand_sequence", "cpm_seed_streams", "cpm_random_fill",
    "cpm_volume_desc_default", "cpm_volume_create", "cpm_volume_update", "cpm_volume_destroy", "cpm_volume_device_data", "cpm_volume_download",
    "cpm_tf_create", "cpm_tf_update", "cpm_tf_destroy", "cpm_uniform_samples_2d", "cpm_directional_light_samples", "cpm_point_light_samples",
    "cpm_light_sample_box_intersection", "cpm_light_sample_mesh_intersection", "cpm_trace", "cpm_grid_desc_default", "cpm_relative_irradiance_scale",
    "cpm_splat", "cpm_splat_selected", "cpm_copy_indexed_photons", "cpm_snapshot_selected_photons", "cpm_sort_pairs", "cpm_sort_keys", "cpm_bin",
    "cpm_gather", "cpm_volume_minmax", "cpm_volume_difference", "cpm_importance_tf", "cpm_photon_importance", "cpm_photon_importance_equal",
    "cpm_reset_importance", "cpm_select_recompute", "cpm_select_changed", "cpm_mix_buffers", "cpm_volume_mix", "cpm_comm_get_unique_id",
    "cpm_comm_create", "cpm_comm_create_all", "cpm_comm_destroy", "cpm_comm_rank", "cpm_comm_size", "cpm_allreduce_grid", "cpm_reduce_grid",
    "cpm_allreduce_grids"
]
#: ... and include/cpm/cpm_ext.h: what this build adds beyond those call sites
EXT_SYMBOLS = [
    "cpm_set_photon_layout", "cpm_get_photon_layout", "cpm_records_describe", "cpm_records_forget", "cpm_trace_lights_order_samples", "cpm_trace_lights",
    "cpm_trace_emitted", "cpm_trace_order_create", "cpm_trace_order_destroy", "cpm_trace_set_order", "cpm_trace_order_update", "cpm_splat_records",
    "cpm_mark_touched_bricks", "cpm_gather_bricks", "cpm_fast_table_entries", "cpm_gather_fast_supported", "cpm_gather_fast_supported_on",
    "cpm_fast_record_capacity", "cpm_bin_fast", "cpm_bin_fast_layout", "cpm_photons_convert", "cpm_gather_fast", "cpm_gather_fast_marked",
    "cpm_volume_step", "cpm_importance_tf_occupancy", "cpm_selection_create", "cpm_selection_destroy", "cpm_selection_begin",
    "cpm_photon_importance_select", "cpm_photon_importance_retrace", "cpm_photon_importance_retrace_lights", "cpm_photon_importance_equal_select",
    "cpm_selection_finish", "cpm_selection_set_occupancy", "cpm_selection_count_device", "cpm_selection_count", "cpm_trace_selected", "cpm_splat_delta",
    "cpm_pinned_alloc", "cpm_pinned_free", "cpm_volume_stream_create", "cpm_volume_stream_destroy", "cpm_volume_stream_prefetch",
    "cpm_volume_stream_acquire", "cpm_volume_stream_stats", "cpm_allreduce_grid_bricks", "cpm_brick_mask_or", "cpm_sparse_reduce_create",
    "cpm_sparse_reduce_destroy", "cpm_sparse_reduce_bricks", "cpm_sparse_reduce_capacity_for", "cpm_allreduce_grid_sparse", "cpm_sparse_reduce_complete",
    "cpm_bricklist_reduce_create", "cpm_bricklist_reduce_destroy", "cpm_bricklist_reduce_bricks", "cpm_bricklist_capacity_for",
    "cpm_bricklist_segment_bytes", "cpm_reduce_grid_bricklists", "cpm_bricklist_reduce_complete", "cpm_bricklist_reduce_open", "cpm_bricklist_pack_grid",
    "cpm_bricklist_reduce_exchange", "cpm_gather_fast_segment", "cpm_bricklist_segment_to_grid", "cpm_comm_send", "cpm_comm_recv",
    "cpm_light_volume_texels", "cpm_gl_available", "cpm_gl_register_buffer", "cpm_gl_acquire", "cpm_gl_release", "cpm_gl_buffer_pointer",
    "cpm_gl_copy_to_buffer", "cpm_gl_unregister"
]
ABI_SYMBOLS = CORE_SYMBOLS + EXT_SYMBOLS
CPM_GL_TEXEL_F32, CPM_GL_TEXEL_F16 = 0, 1


CPM_MIX_F32, CPM_MIX_U16X2 = 0, 1


class TraceOrder:
    """The launch order of a trace over n_light_samples samples, fed by the launches' own costs (cpm_trace_order)."""

    def __init__(self, ctx, n_light_samples: int):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.lib.cpm_trace_order_create(ctx.h, int(n_light_samples), C.byref(h)))
        self.h = h
        self.n_light_samples = int(n_light_samples)

    def update(self):
        self.ctx._check(self.ctx.lib.cpm_trace_order_update(self.ctx.h, self.h, self.ctx._stream()))

    def read(self):
        """(order table, costs gathered since the last update, launches counted) as numpy -- test hook, synchronises."""
        import numpy as np
        n = (self.n_light_samples + 255) // 256
        order, cost = np.zeros(n, np.uint32), np.zeros(n + 1, np.uint32)
        self.ctx._check(self.ctx.lib.cpm_debug_trace_order_read(self.ctx.h, self.h, order.ctypes.data, cost.ctypes.data))
        return order, cost[:n], int(cost[n])

    def write(self, table):
        """Replace the table (a permutation of the chunks) -- measurement hook, synchronises."""
        import numpy as np
        t = np.ascontiguousarray(table, np.uint32)
        self.ctx._check(self.ctx.lib.cpm_debug_trace_order_write(self.ctx.h, self.h, t.ctypes.data))

    def close(self):
        if self.h:
            self.ctx.lib.cpm_trace_order_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GLResource:
    """A registered OpenGL buffer object (cpm_gl_resource)."""

    def __init__(self, ctx, h):
        self.ctx, self.h = ctx, h

    def pointer(self):
        """(device address, bytes) of an acquired buffer."""
        p, n = C.c_void_p(), C.c_size_t(0)
        self.ctx._check(self.ctx.lib.cpm_gl_buffer_pointer(self.ctx.h, self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        if self.h:
            self.ctx.lib.cpm_gl_unregister(self.ctx.h, self.h)
            self.h = None


class VolumeDesc(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3), ("dtype", C.c_int32), ("format_scaling", C.c_float),
                ("format_offset", C.c_float), ("texture_to_index", C.c_float * 16),
                ("index_to_texture", C.c_float * 16)]


class GridDesc(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3), ("channels", C.c_int32),
                ("texture_to_index", C.c_float * 16), ("index_to_texture", C.c_float * 16)]


class TraceParams(C.Structure):
    _fields_ = [("material", C.c_float * 4), ("step_size", C.c_float), ("photon_offset", C.c_int32),
                ("n_light_samples", C.c_int32), ("max_interactions", C.c_int32),
                ("total_photons", C.c_int32), ("shading_type", C.c_int32), ("flags", C.c_int32),
                ("iteration", C.c_int32), ("batch", C.c_int32)]


class LightSpan(C.Structure):
    """cpm_light_span: one light of a cpm_trace_lights launch."""
    _fields_ = [("light_samples8", C.c_void_p), ("isect2", C.c_void_p), ("n_light_samples", C.c_int32), ("photon_offset", C.c_int32)]


CPM_EMIT_DIRECTIONAL, CPM_EMIT_POINT = 0, 1


class SparseReduceInfo(C.Structure):
    """cpm_sparse_reduce_info"""
    _fields_ = [("ticket", C.c_uint64), ("n_bricks", C.c_uint32), ("n_union", C.c_uint32), ("capacity", C.c_uint32), ("mode", C.c_int),
                ("reduce_bytes", C.c_uint64), ("dense_bytes", C.c_uint64)]


class BricklistInfo(C.Structure):
    """cpm_bricklist_info"""
    _fields_ = [("ticket", C.c_uint64), ("n_bricks", C.c_uint32), ("n_own", C.c_uint32), ("capacity", C.c_uint32), ("resent", C.c_int32),
                ("sent_bytes", C.c_uint64), ("received_bytes", C.c_uint64), ("dense_bytes", C.c_uint64), ("listed_bricks", C.c_uint32)]


class VolumeStreamInfo(C.Structure):
    """cpm_volume_stream_info"""
    _fields_ = [("uploads", C.c_uint64), ("hits", C.c_uint64), ("uploads_at_acquire", C.c_uint64), ("bytes_uploaded", C.c_uint64),
                ("bytes_per_step", C.c_uint64), ("uploads_timed", C.c_uint64), ("upload_ms_total", C.c_double)]


class BricklistSegment(C.Structure):
    """cpm_bricklist_segment: where a ticket's brick list lies on a sender (segment == NULL at the root: it gathers into its grid)."""
    _fields_ = [("segment", C.c_void_p), ("capacity", C.c_uint32), ("room", C.c_uint32), ("ticket", C.c_uint32), ("channels", C.c_uint32),
                ("control", C.c_void_p), ("mailbox", C.c_void_p)]


class EmitterDesc(C.Structure):
    """cpm_emitter_desc: the light of cpm_trace_emitted (lattice + directional light plane or point light)."""
    _fields_ = [("kind", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("first_sample", C.c_int32),
                ("radiance", C.c_float * 4), ("direction_or_position", C.c_float * 4),
                ("plane_origin", C.c_float * 4), ("tangent_u", C.c_float * 4), ("tangent_v", C.c_float * 4),
                ("plane_area", C.c_float)]


def directional_emitter(nx, ny, radiance, direction, origin, u, v, area, first_sample=0) -> EmitterDesc:
    return EmitterDesc(CPM_EMIT_DIRECTIONAL, nx, ny, first_sample, _f4(radiance), _f4(direction), _f4(origin), _f4(u),
                       _f4(v), float(area))


def point_emitter(nx, ny, radiance, position, first_sample=0) -> EmitterDesc:
    z = _f4((0, 0, 0))
    return EmitterDesc(CPM_EMIT_POINT, nx, ny, first_sample, _f4(radiance), _f4(position), z, _f4((0, 0, 0)), _f4((0, 0, 0)), 0.0)


class CpmError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libcpm_hip status {status}: {message}")
        self.status = status


_lib = None


def load_library() -> C.CDLL:
    """Load libcpm_hip.so from the package directory; fail loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    lib_path = LIB_PATH
    if os.environ.get("CPM_LIB"):  # tuning experiments (tools/): another build of the same library
        lib_path = Path(os.environ["CPM_LIB"])
    if not lib_path.exists():
        raise ImportError(
            f"{lib_path} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback.")
    lib = C.CDLL(str(lib_path))
    vp, i32, f32, u32, u64, sz = C.c_void_p, C.c_int32, C.c_float, C.c_uint32, C.c_uint64, C.c_size_t
    P = C.POINTER
    sig = {
        "cpm_create": (i32, [i32, P(vp)]),
        "cpm_destroy": (None, [vp]),
        "cpm_last_error_string": (C.c_char_p, [vp]),
        "cpm_abi_version": (i32, []),
        "cpm_glibc_rand_sequence": (None, [u32, vp, sz]),
        "cpm_seed_streams": (i32, [vp, vp, sz, u64, vp]),
        "cpm_random_fill": (i32, [vp, vp, sz, i32, vp, vp]),
        "cpm_volume_desc_default": (None, [P(VolumeDesc), P(i32 * 3), i32]),
        "cpm_volume_create": (i32, [vp, P(VolumeDesc), vp, i32, vp, P(vp)]),
        "cpm_volume_update": (i32, [vp, vp, vp, i32, vp]),
        "cpm_volume_destroy": (None, [vp, vp]),
        "cpm_tf_create": (i32, [vp, vp, i32, i32, vp, P(vp)]),
        "cpm_tf_update": (i32, [vp, vp, vp, i32, vp]),
        "cpm_tf_destroy": (None, [vp, vp]),
        "cpm_uniform_samples_2d": (i32, [vp, i32, i32, vp, vp]),
        "cpm_directional_light_samples": (i32, [vp, vp, i32, P(f32 * 4), P(f32 * 4), P(f32 * 4), P(f32 * 4), P(f32 * 4), f32, vp, vp]),
        "cpm_point_light_samples": (i32, [vp, vp, i32, P(f32 * 4), P(f32 * 4), vp, vp]),
        "cpm_light_sample_box_intersection": (i32, [vp, vp, i32, P(f32 * 8), vp, vp]),
        "cpm_light_sample_mesh_intersection": (i32, [vp, vp, vp, i32, vp, i32, vp, vp]),
        "cpm_trace": (i32, [vp, vp, vp, vp, P(f32 * 8), P(TraceParams), vp, vp, vp, i32, vp, vp, vp]),
        "cpm_trace_lights": (i32, [vp, vp, vp, vp, P(f32 * 8), P(TraceParams), vp, i32, vp, vp, vp]),
        "cpm_trace_lights_order_samples": (i32, [vp, i32]),
        "cpm_trace_emitted": (i32, [vp, vp, vp, vp, P(f32 * 8), P(TraceParams), P(EmitterDesc), vp, i32, vp, vp, vp]),
        "cpm_grid_desc_default": (None, [P(GridDesc), P(i32 * 3), i32]),
        "cpm_relative_irradiance_scale": (f32, [C.c_double, C.c_double]),
        "cpm_splat": (i32, [vp, vp, i32, P(GridDesc), f32, f32, vp, vp]),
        "cpm_splat_selected": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, f32, i32, i32, vp, vp]),
        "cpm_copy_indexed_photons": (i32, [vp, vp, vp, i32, f32, i32, i32, vp, i32, vp]),
        "cpm_snapshot_selected_photons": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
        "cpm_sort_pairs": (i32, [vp, vp, vp, sz, i32, vp]),
        "cpm_sort_keys": (i32, [vp, vp, sz, i32, vp]),
        "cpm_bin": (i32, [vp, vp, i32, P(GridDesc), vp, vp, vp, vp]),
        "cpm_gather": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, i32, vp, vp]),
        "cpm_mark_touched_bricks": (i32, [vp, vp, vp, i32, i32, i32, P(GridDesc), f32, vp, vp]),
        "cpm_gather_bricks": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, vp, vp, vp]),
        "cpm_fast_table_entries": (sz, [P(GridDesc), i32]),
        "cpm_gather_fast_supported": (i32, [P(GridDesc), f32]),
        "cpm_fast_record_capacity": (sz, [P(GridDesc), i32, f32]),
        "cpm_bin_fast": (i32, [vp, vp, i32, P(GridDesc), f32, vp, vp, vp]),
        "cpm_bin_fast_layout": (i32, [vp, vp, i32, i32, P(GridDesc), f32, vp, vp, vp]),
        "cpm_photons_convert": (i32, [vp, vp, i32, vp, i32, C.c_size_t, vp]),
        "cpm_set_photon_layout": (i32, [vp, i32]),
        "cpm_records_describe": (i32, [vp, vp, i32, C.c_size_t]),
        "cpm_records_forget": (i32, [vp, vp]),
        "cpm_get_photon_layout": (i32, [vp]),
        "cpm_splat_records": (i32, [vp, vp, i32, i32, P(GridDesc), f32, f32, vp, vp]),
        "cpm_gather_fast": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, i32, vp, vp]),
        "cpm_gather_fast_marked": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, i32, vp, vp, vp]),
        "cpm_volume_minmax": (i32, [vp, vp, i32, vp, vp]),
        "cpm_volume_difference": (i32, [vp, vp, vp, i32, vp, vp]),
        "cpm_volume_step": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "cpm_importance_tf": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp]),
        "cpm_importance_tf_occupancy": (i32, [vp, vp, vp, vp, i32, vp, vp, i32, vp, vp, vp]),
        "cpm_selection_set_occupancy": (i32, [vp, vp, vp, vp]),
        "cpm_photon_importance": (i32, [vp, vp, P(i32 * 3), P(f32 * 3), P(f32 * 16), vp, i32, vp, vp, i32, i32, i32, i32, vp, vp]),
        "cpm_photon_importance_equal": (i32, [vp, i32, i32, i32, i32, vp, vp]),
        "cpm_reset_importance": (i32, [vp, vp, sz, sz, vp]),
        "cpm_select_recompute": (i32, [vp, vp, sz, vp, vp, vp]),
        "cpm_select_changed": (i32, [vp, vp, sz, vp, vp, vp]),
        "cpm_selection_create": (i32, [vp, sz, P(vp)]),
        "cpm_selection_destroy": (None, [vp, vp]),
        "cpm_selection_begin": (i32, [vp, vp]),
        "cpm_photon_importance_select": (i32, [vp, vp, vp, P(i32 * 3), P(f32 * 3), P(f32 * 16), vp, i32, vp, vp, i32, i32, i32, i32, vp, vp]),
        "cpm_photon_importance_equal_select": (i32, [vp, vp, i32, i32, i32, i32, vp, vp]),
        "cpm_photon_importance_retrace": (i32, [vp, vp, vp, P(i32 * 3), P(f32 * 3), P(f32 * 16), vp, vp, vp, P(f32 * 8), P(TraceParams), vp, vp, i32,
                                          vp, vp, vp, vp, vp]),
        "cpm_photon_importance_retrace_lights": (i32, [vp, vp, vp, P(i32 * 3), P(f32 * 3), P(f32 * 16), vp, vp, vp, P(f32 * 8), P(TraceParams), vp, i32, i32,
                                                 vp, vp, vp, vp, vp]),
        "cpm_selection_finish": (i32, [vp, vp, vp, vp]),
        "cpm_selection_count_device": (vp, [vp]),
        "cpm_selection_count": (i32, [vp, vp, P(i32)]),
        "cpm_trace_selected": (i32, [vp, vp, vp, vp, P(f32 * 8), P(TraceParams), vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]),
        "cpm_splat_delta": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, P(GridDesc), f32, f32, i32, i32, vp, vp, vp]),
        "cpm_mix_buffers": (i32, [vp, vp, vp, f32, sz, i32, vp, vp]),
        "cpm_volume_mix": (i32, [vp, vp, vp, f32, vp, vp]),
        "cpm_comm_get_unique_id": (i32, [vp, vp]),
        "cpm_comm_create": (i32, [vp, vp, i32, i32, P(vp)]),
        "cpm_comm_create_all": (i32, [P(vp), i32, P(vp)]),
        "cpm_comm_destroy": (None, [vp]),
        "cpm_comm_rank": (i32, [vp]),
        "cpm_comm_size": (i32, [vp]),
        "cpm_allreduce_grid": (i32, [vp, vp, vp, vp, sz, vp]),
        "cpm_reduce_grid": (i32, [vp, vp, vp, vp, sz, i32, vp]),
        "cpm_allreduce_grids": (i32, [P(vp), P(vp), P(vp), sz, P(vp), i32]),
        "cpm_allreduce_grid_bricks": (i32, [vp, vp, vp, vp, P(GridDesc), vp, P(u32), vp]),
        "cpm_sparse_reduce_create": (i32, [vp, vp, P(GridDesc), P(vp)]),
        "cpm_sparse_reduce_destroy": (None, [vp]),
        "cpm_sparse_reduce_bricks": (u32, [vp]),
        "cpm_sparse_reduce_capacity_for": (u32, [u32, C.c_longlong]),
        "cpm_allreduce_grid_sparse": (i32, [vp, vp, vp, vp, vp, i32, i32, u32, P(C.c_uint64), vp]),
        "cpm_sparse_reduce_complete": (i32, [vp, vp, C.c_uint64, vp, P(SparseReduceInfo)]),
        "cpm_brick_mask_or": (i32, [vp, vp, vp, sz, vp]),
        "cpm_bricklist_reduce_create": (i32, [vp, vp, P(GridDesc), i32, P(vp)]),
        "cpm_bricklist_reduce_destroy": (None, [vp]),
        "cpm_bricklist_reduce_bricks": (u32, [vp]),
        "cpm_bricklist_capacity_for": (u32, [u32, C.c_longlong]),
        "cpm_bricklist_segment_bytes": (C.c_uint64, [u32, i32]),
        "cpm_reduce_grid_bricklists": (i32, [vp, vp, vp, vp, P(C.c_uint64), vp]),
        "cpm_bricklist_reduce_complete": (i32, [vp, vp, C.c_uint64, vp, P(BricklistInfo)]),
        "cpm_bricklist_reduce_open": (i32, [vp, vp, P(C.c_uint64), P(BricklistSegment)]),
        "cpm_bricklist_pack_grid": (i32, [vp, vp, C.c_uint64, vp, vp, vp]),
        "cpm_bricklist_reduce_exchange": (i32, [vp, vp, C.c_uint64, vp, vp]),
        "cpm_gather_fast_segment": (i32, [vp, vp, vp, i32, P(GridDesc), f32, f32, P(BricklistSegment), vp]),
        "cpm_bricklist_segment_to_grid": (i32, [vp, P(BricklistSegment), P(GridDesc), vp, vp]),
        "cpm_comm_send": (i32, [vp, vp, vp, C.c_size_t, i32, vp]),
        "cpm_comm_recv": (i32, [vp, vp, vp, C.c_size_t, i32, vp]),
        "cpm_gather_fast_supported_on": (i32, [vp, P(GridDesc), f32]),
        "cpm_pinned_alloc": (i32, [vp, C.c_size_t, P(vp)]),
        "cpm_pinned_free": (None, [vp, vp]),
        "cpm_volume_stream_create": (i32, [vp, P(VolumeDesc), i32, P(vp)]),
        "cpm_volume_stream_destroy": (None, [vp, vp]),
        "cpm_volume_stream_prefetch": (i32, [vp, vp, C.c_uint64, vp, vp]),
        "cpm_volume_stream_acquire": (i32, [vp, vp, C.c_uint64, vp, vp, P(vp)]),
        "cpm_volume_stream_stats": (i32, [vp, vp, P(VolumeStreamInfo)]),
        "cpm_gl_available": (i32, [vp]),
        "cpm_gl_register_buffer": (i32, [vp, u32, i32, P(vp)]),
        "cpm_light_volume_texels": (i32, [vp, vp, sz, i32, vp, vp]),
        "cpm_trace_order_create": (i32, [vp, i32, P(vp)]),
        "cpm_trace_order_destroy": (None, [vp, vp]),
        "cpm_trace_set_order": (i32, [vp, vp, i32]),
        "cpm_trace_order_update": (i32, [vp, vp, vp]),
        "cpm_gl_acquire": (i32, [vp, P(vp), i32, vp]),
        "cpm_gl_release": (i32, [vp, P(vp), i32, vp]),
        "cpm_gl_buffer_pointer": (i32, [vp, vp, P(vp), P(sz)]),
        "cpm_gl_copy_to_buffer": (i32, [vp, vp, sz, i32, vp, vp]),
        "cpm_gl_unregister": (None, [vp, vp]),
        "cpm_volume_device_data": (vp, [vp, P(sz)]),
        "cpm_volume_download": (i32, [vp, vp, vp, vp]),
        # include/cpm/cpm_profile.h (measurement hooks)
        "cpm_debug_trace_order_read": (i32, [vp, vp, vp, vp]),
        "cpm_debug_trace_order_write": (i32, [vp, vp, vp]),
        "cpm_debug_set_step_counter": (None, [vp, vp]),
        "cpm_debug_set_gather_stamps": (None, [vp, vp]),
        "cpm_debug_force_voxel_gather": (None, [vp, i32]),
        "cpm_debug_set_gather_coop": (None, [vp, i32]),
        "cpm_debug_set_brick_streaming": (None, [vp, i32]),
        "cpm_debug_set_bin_fused": (None, [vp, i32]),
        "cpm_debug_set_select_partition": (None, [vp, i32]),
        "cpm_debug_set_sort_mode": (None, [vp, i32]),
        "cpm_debug_set_sort_items": (None, [vp, i32]),
        "cpm_debug_set_stream_wg_per_cu": (None, [vp, i32]),
        "cpm_debug_fail_next_select": (None, [vp, i32]),
        "cpm_debug_pack_grid_segment": (i32, [vp, P(BricklistSegment), P(GridDesc), vp, vp, vp]),
        "cpm_debug_root_add_segments": (i32, [vp, P(BricklistSegment), i32, P(GridDesc), vp, vp, vp]),
        "cpm_profile_enable": (None, [vp, i32]),
        "cpm_profile_reset": (None, [vp]),
        "cpm_profile_collect": (i32, [vp]),
        "cpm_profile_name": (C.c_char_p, [vp, i32]),
        "cpm_profile_total_ms": (C.c_double, [vp, i32]),
        "cpm_profile_calls": (C.c_long, [vp, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_volume_desc(dims, dtype) -> VolumeDesc:
    d = VolumeDesc()
    load_library().cpm_volume_desc_default(C.byref(d), C.byref((C.c_int32 * 3)(*dims)), dtype)
    return d


def default_grid_desc(dims, channels=1) -> GridDesc:
    d = GridDesc()
    load_library().cpm_grid_desc_default(C.byref(d), C.byref((C.c_int32 * 3)(*dims)), channels)
    return d


def relative_irradiance_scale(radius_relative_to_scene: float, n_photons: float) -> float:
    return float(load_library().cpm_relative_irradiance_scale(radius_relative_to_scene, n_photons))


def glibc_rand_sequence(seed: int, n: int):
    import numpy as np
    out = np.zeros(n, dtype=np.uint32)
    load_library().cpm_glibc_rand_sequence(seed, out.ctypes.data, n)
    return out


def _f4(v):
    v = list(v) + [0.0] * (4 - len(v))
    return (C.c_float * 4)(*v[:4])


_TORCH_DTYPES = None


def _dtype_code(t):
    import torch
    return {torch.uint8: CPM_U8, torch.uint16: CPM_U16, torch.int16: CPM_U16, torch.float32: CPM_F32}[t.dtype]


class Context:
    """One libcpm_hip context on one GPU.  Methods take torch tensors living on that GPU."""

    def __init__(self, device: int = 0):
        import torch
        self.lib = load_library()
        self.device_index = device
        h = C.c_void_p()
        rc = self.lib.cpm_create(device, C.byref(h))
        if rc != CPM_OK:
            raise CpmError(rc, self.lib.cpm_last_error_string(None).decode())
        self.h = h
        self.torch = torch
        self.device = torch.device("cuda", device)

    def close(self):
        if getattr(self, "h", None):
            self.lib.cpm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t, dtype=None):
        if t is None:
            return None
        if not t.is_cuda or t.device.index != self.device_index:
            raise ValueError("tensor is not on this context's GPU")
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise ValueError(f"expected dtype {dtype}, got {t.dtype}")
        return C.c_void_p(t.data_ptr())

    def _check(self, rc):
        if rc != CPM_OK:
            raise CpmError(rc, self.lib.cpm_last_error_string(self.h).decode())

    # -- measurement hooks
    def profile_enable(self, on=True):
        self.lib.cpm_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.lib.cpm_profile_reset(self.h)

    def profile_collect(self):
        """{kernel name: (total ms, calls)} of every launch since the last reset."""
        n = self.lib.cpm_profile_collect(self.h)
        return {self.lib.cpm_profile_name(self.h, i).decode(): (self.lib.cpm_profile_total_ms(self.h, i),
                                                                 self.lib.cpm_profile_calls(self.h, i)) for i in range(n)}

    def set_step_counter(self, counter_tensor):
        """int64[1] device tensor that trace launches add their Woodcock iteration counts to (None = off)."""
        self.lib.cpm_debug_set_step_counter(self.h, self._ptr(counter_tensor))

    # -- RNG
    def seed_streams(self, state, gap=1 << 40):
        """state: uint32 [n, 2] with state[:, 0] = per-stream base offsets; in place."""
        self._check(self.lib.cpm_seed_streams(self.h, self._ptr(state), state.shape[0], gap, self._stream()))

    def random_fill(self, state, draws):
        out = self.torch.empty((draws, state.shape[0]), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_random_fill(self.h, self._ptr(state), state.shape[0], draws, self._ptr(out), self._stream()))
        return out

    # -- volume / tf
    def volume_create(self, voxels, desc: VolumeDesc | None = None):
        """voxels: torch tensor [z, y, x] (u8 / u16-as-int16|uint16 / f32) on the GPU, or a numpy array on the host."""
        import numpy as np
        if isinstance(voxels, np.ndarray):
            code = {np.dtype(np.uint8): CPM_U8, np.dtype(np.uint16): CPM_U16, np.dtype(np.float32): CPM_F32}[voxels.dtype]
            dims = voxels.shape[::-1]
            voxels = np.ascontiguousarray(voxels)
            ptr, is_dev = C.c_void_p(voxels.ctypes.data), 0
        else:
            code = _dtype_code(voxels)
            dims = tuple(voxels.shape[::-1])
            ptr, is_dev = self._ptr(voxels), 1
        if desc is None:
            desc = default_volume_desc(dims, code)
        h = C.c_void_p()
        self._check(self.lib.cpm_volume_create(self.h, C.byref(desc), ptr, is_dev, self._stream(), C.byref(h)))
        return Volume(self, h, desc)

    def tf_create(self, rgba):
        """rgba: [width, 4] float32, torch (GPU) or numpy (host)."""
        import numpy as np
        if isinstance(rgba, np.ndarray):
            rgba = np.ascontiguousarray(rgba, dtype=np.float32)
            ptr, is_dev, width = C.c_void_p(rgba.ctypes.data), 0, rgba.shape[0]
        else:
            ptr, is_dev, width = self._ptr(rgba, self.torch.float32), 1, rgba.shape[0]
        h = C.c_void_p()
        self._check(self.lib.cpm_tf_create(self.h, ptr, width, is_dev, self._stream(), C.byref(h)))
        return TransferFunction(self, h, width)

    # -- emission
    def uniform_samples_2d(self, nx, ny):
        out = self.torch.empty((nx * ny, 4), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_uniform_samples_2d(self.h, nx, ny, self._ptr(out), self._stream()))
        return out

    def directional_light_samples(self, samples, radiance, direction, origin, tangent_u, tangent_v, area):
        n = samples.shape[0]
        out = self.torch.empty((n, 8), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_directional_light_samples(
            self.h, self._ptr(samples), n, C.byref(_f4(radiance)), C.byref(_f4(direction)), C.byref(_f4(origin)),
            C.byref(_f4(tangent_u)), C.byref(_f4(tangent_v)), float(area), self._ptr(out), self._stream()))
        return out

    def point_light_samples(self, samples, radiance, position):
        n = samples.shape[0]
        out = self.torch.empty((n, 8), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_point_light_samples(self.h, self._ptr(samples), n, C.byref(_f4(radiance)),
                                                     C.byref(_f4(position)), self._ptr(out), self._stream()))
        return out

    def light_sample_box_intersection(self, light_samples, aabb):
        n = light_samples.shape[0]
        out = self.torch.empty((n, 2), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_light_sample_box_intersection(self.h, self._ptr(light_samples), n,
                                                               C.byref((C.c_float * 8)(*aabb)), self._ptr(out), self._stream()))
        return out

    def light_sample_mesh_intersection(self, vertices, indices, light_samples):
        n = light_samples.shape[0]
        out = self.torch.empty((n, 2), dtype=self.torch.float32, device=self.device)
        self._check(self.lib.cpm_light_sample_mesh_intersection(self.h, self._ptr(vertices), self._ptr(indices),
                                                                indices.numel(), self._ptr(light_samples), n,
                                                                self._ptr(out), self._stream()))
        return out

    # -- trace
    def trace(self, vol, tf, aabb, params: TraceParams, light_samples, isect, rng_state, photons,
              recompute_indices=None, n_recompute=0, tf_scattering=None):
        self._check(self.lib.cpm_trace(
            self.h, vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), self._ptr(light_samples), self._ptr(isect),
            self._ptr(recompute_indices), n_recompute, self._ptr(rng_state), self._ptr(photons), self._stream()))

    def light_spans(self, lights):
        """lights: [(light_samples, isect, n_light_samples, photon_offset), ...] -> a cpm_light_span array."""
        arr = (LightSpan * len(lights))()
        for a, (ls, isect, n, off) in zip(arr, lights):
            a.light_samples8, a.isect2, a.n_light_samples, a.photon_offset = self._ptr(ls), self._ptr(isect), n, off
        return arr

    def trace_lights_order_samples(self, spans) -> int:
        """The number of samples a cpm_trace_order for a cpm_trace_lights launch over `spans` is created for."""
        return int(self.lib.cpm_trace_lights_order_samples(C.cast(spans, C.c_void_p), len(spans)))

    def trace_lights(self, vol, tf, aabb, params: TraceParams, spans, rng_state, photons, tf_scattering=None):
        """cpm_trace for several lights in one launch (spans: light_spans(...))."""
        self._check(self.lib.cpm_trace_lights(
            self.h, vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), C.cast(spans, C.c_void_p), len(spans), self._ptr(rng_state), self._ptr(photons),
            self._stream()))

    def trace_emitted(self, vol, tf, aabb, params: TraceParams, emitter: EmitterDesc, rng_state, photons,
                      recompute_indices=None, n_recompute=0, tf_scattering=None):
        """cpm_trace with the emission chain evaluated in the tracer (no light-sample / intersection buffers read)."""
        self._check(self.lib.cpm_trace_emitted(
            self.h, vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), C.byref(emitter),
            self._ptr(recompute_indices), n_recompute, self._ptr(rng_state), self._ptr(photons), self._stream()))

    # -- light volume
    def splat(self, photons, total_photons, grid: GridDesc, radius, scale, out):
        self._check(self.lib.cpm_splat(self.h, self._ptr(photons), total_photons, C.byref(grid), radius, scale,
                                       self._ptr(out), self._stream()))

    def splat_selected(self, photons, indices, n_indices, grid, radius, scale, multiplier, n_photons, n_interactions, out):
        self._check(self.lib.cpm_splat_selected(self.h, self._ptr(photons), self._ptr(indices), n_indices, C.byref(grid),
                                                radius, scale, multiplier, n_photons, n_interactions, self._ptr(out),
                                                self._stream()))

    def copy_indexed_photons(self, photons, indices, n_indices, multiplier, n_photons, n_interactions, aligned, out_offset=0):
        self._check(self.lib.cpm_copy_indexed_photons(self.h, self._ptr(photons), self._ptr(indices), n_indices, multiplier,
                                                      n_photons, n_interactions, self._ptr(aligned), out_offset, self._stream()))

    def snapshot_selected_photons(self, photons, indices, n_indices, n_photons, n_interactions, snapshot):
        self._check(self.lib.cpm_snapshot_selected_photons(self.h, self._ptr(photons), self._ptr(indices), n_indices, n_photons,
                                                           n_interactions, self._ptr(snapshot), self._stream()))

    def sort_pairs(self, keys, values, key_bits=0):
        self._check(self.lib.cpm_sort_pairs(self.h, self._ptr(keys), self._ptr(values), keys.numel(), key_bits, self._stream()))

    def sort_keys(self, keys, key_bits=0):
        self._check(self.lib.cpm_sort_keys(self.h, self._ptr(keys), keys.numel(), key_bits, self._stream()))

    def bin(self, photons, n, grid: GridDesc, order, cell_start, sorted_pos_power):
        self._check(self.lib.cpm_bin(self.h, self._ptr(photons), n, C.byref(grid), self._ptr(order), self._ptr(cell_start),
                                     self._ptr(sorted_pos_power), self._stream()))

    def gather(self, sorted_pos_power, cell_start, n, grid, radius, scale, out, accumulate=False):
        self._check(self.lib.cpm_gather(self.h, self._ptr(sorted_pos_power), self._ptr(cell_start), n, C.byref(grid), radius,
                                        scale, int(accumulate), self._ptr(out), self._stream()))

    def fast_table_entries(self, grid: GridDesc, n: int) -> int:
        return int(self.lib.cpm_fast_table_entries(C.byref(grid), n))

    def gather_fast_supported(self, grid: GridDesc, radius: float) -> bool:
        return bool(self.lib.cpm_gather_fast_supported(C.byref(grid), radius))

    def fast_record_capacity(self, grid: GridDesc, n: int, radius: float) -> int:
        return int(self.lib.cpm_fast_record_capacity(C.byref(grid), n, radius))

    def bin_fast(self, photons, n, grid: GridDesc, radius, brick_table, sorted_pos_power, layout=None):
        """layout: None (cpm_bin_fast: the context's layout), CPM_PHOTONS_INTERLEAVED (float8 records) or CPM_PHOTONS_PLANAR (what a
        CPM_TRACE_PHOTONS_PLANAR trace wrote)."""
        if layout is None:
            self._check(self.lib.cpm_bin_fast(self.h, self._ptr(photons), n, C.byref(grid), radius, self._ptr(brick_table),
                                              self._ptr(sorted_pos_power), self._stream()))
        else:
            self._check(self.lib.cpm_bin_fast_layout(self.h, self._ptr(photons), layout, n, C.byref(grid), radius, self._ptr(brick_table),
                                                     self._ptr(sorted_pos_power), self._stream()))

    def set_photon_layout(self, layout):
        """CPM_PHOTONS_INTERLEAVED / CPM_PHOTONS_PLANAR for every N * I record buffer handed to this context (cpm_set_photon_layout)."""
        self._check(self.lib.cpm_set_photon_layout(self.h, int(layout)))

    def photon_layout(self):
        return int(self.lib.cpm_get_photon_layout(self.h))

    def records_describe(self, photons, layout, n_records=None):
        """This buffer holds n_records (default: its rows) records in `layout`, whatever the context's default (cpm_records_describe)."""
        n = int(photons.shape[0] if n_records is None else n_records)
        self._check(self.lib.cpm_records_describe(self.h, self._ptr(photons), int(layout), n))

    def records_forget(self, photons):
        self._check(self.lib.cpm_records_forget(self.h, self._ptr(photons)))

    def splat_records(self, photons, n_records, total_photons, grid: GridDesc, radius, scale, out):
        self._check(self.lib.cpm_splat_records(self.h, self._ptr(photons), n_records, total_photons, C.byref(grid), radius, scale,
                                               self._ptr(out), self._stream()))

    def photons_convert(self, src, src_layout, dst, dst_layout, n_records):
        self._check(self.lib.cpm_photons_convert(self.h, self._ptr(src), src_layout, self._ptr(dst), dst_layout, n_records, self._stream()))

    def gather_fast(self, sorted_pos_power, brick_table, n, grid, radius, scale, out, accumulate=False, nonzero_bricks=None):
        """nonzero_bricks (uint8, one per 4x4x4 brick): also written -- 1 where the volume is not zero (cpm_gather_fast_marked)."""
        if nonzero_bricks is None:
            self._check(self.lib.cpm_gather_fast(self.h, self._ptr(sorted_pos_power), self._ptr(brick_table), n, C.byref(grid), radius, scale,
                                                 int(accumulate), self._ptr(out), self._stream()))
        else:
            self._check(self.lib.cpm_gather_fast_marked(self.h, self._ptr(sorted_pos_power), self._ptr(brick_table), n, C.byref(grid), radius, scale,
                                                        int(accumulate), self._ptr(out), self._ptr(nonzero_bricks), self._stream()))

    def gather_fast_segment(self, sorted_pos_power, brick_table, n, grid, radius, scale, segment: "BricklistSegment"):
        """cpm_gather_fast whose output is a brick-list segment (BricklistReduce.open): the non-zero 4x4x4 bricks alone, no grid."""
        self._check(self.lib.cpm_gather_fast_segment(self.h, self._ptr(sorted_pos_power), self._ptr(brick_table), n, C.byref(grid), radius, scale,
                                                     C.byref(segment), self._stream()))

    def bricklist_segment_to_grid(self, segment: "BricklistSegment", grid, out):
        """out += the segment's bricks (same device)."""
        self._check(self.lib.cpm_bricklist_segment_to_grid(self.h, C.byref(segment), C.byref(grid), self._ptr(out), self._stream()))

    def gather_fast_supported_on(self, grid, radius) -> bool:
        return bool(self.lib.cpm_gather_fast_supported_on(self.h, C.byref(grid), radius))

    def comm_send(self, comm: "Comm", buf, nbytes: int, peer: int):
        self._check(self.lib.cpm_comm_send(self.h, comm.h, self._ptr(buf), nbytes, peer, self._stream()))

    def comm_recv(self, comm: "Comm", buf, nbytes: int, peer: int):
        self._check(self.lib.cpm_comm_recv(self.h, comm.h, self._ptr(buf), nbytes, peer, self._stream()))

    def debug_pack_grid_segment(self, segment: "BricklistSegment", grid_desc, grid, nonzero_bricks=None):
        """(measurement hook, cpm_profile.h) the sender's pack launch of cpm_bricklist_pack_grid into a caller-made segment."""
        self._check(self.lib.cpm_debug_pack_grid_segment(self.h, C.byref(segment), C.byref(grid_desc), self._ptr(grid),
                                                         self._ptr(nonzero_bricks) if nonzero_bricks is not None else None, self._stream()))

    def debug_root_add_segments(self, segments, grid_desc, grid, slot_of):
        """(measurement hook) the root's two launches of cpm_bricklist_reduce_exchange over caller-made segments: grid += segments."""
        arr = (BricklistSegment * len(segments))(*segments)
        self._check(self.lib.cpm_debug_root_add_segments(self.h, arr, len(segments), C.byref(grid_desc), self._ptr(grid), self._ptr(slot_of), self._stream()))

    def mark_touched_bricks(self, photons, indices, n_indices, n_photons, n_interactions, grid, radius, brick_mask):
        self._check(self.lib.cpm_mark_touched_bricks(self.h, self._ptr(photons), self._ptr(indices), n_indices, n_photons,
                                                     n_interactions, C.byref(grid), radius, self._ptr(brick_mask), self._stream()))

    def gather_bricks(self, sorted_pos_power, cell_start, n, grid, radius, scale, brick_mask, out):
        self._check(self.lib.cpm_gather_bricks(self.h, self._ptr(sorted_pos_power), self._ptr(cell_start), n, C.byref(grid), radius,
                                               scale, self._ptr(brick_mask), self._ptr(out), self._stream()))

    # -- multi-GPU: the one exchange step (RCCL through the C-ABI)
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * 128)()
        self._check(self.lib.cpm_comm_get_unique_id(self.h, buf))
        return bytes(buf)

    def comm_create(self, unique_id: bytes, rank: int, n_ranks: int) -> "Comm":
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self.lib.cpm_comm_create(self.h, buf, rank, n_ranks, C.byref(h)))
        return Comm(self, h)

    def allreduce_grid(self, comm: "Comm", grid, out=None):
        """out (default: in place) = sum over ranks of grid."""
        recv = grid if out is None else out
        self._check(self.lib.cpm_allreduce_grid(self.h, comm.h, self._ptr(grid), self._ptr(recv), grid.numel(), self._stream()))

    def reduce_grid(self, comm: "Comm", send, recv, root: int):
        self._check(self.lib.cpm_reduce_grid(self.h, comm.h, self._ptr(send), self._ptr(recv) if recv is not None else None,
                                             send.numel(), root, self._stream()))

    def allreduce_grid_bricks(self, comm: "Comm", partial, total, grid: GridDesc, brick_mask) -> int:
        n_union = C.c_uint32(0)
        self._check(self.lib.cpm_allreduce_grid_bricks(self.h, comm.h, self._ptr(partial), self._ptr(total), C.byref(grid),
                                                       self._ptr(brick_mask), C.byref(n_union), self._stream()))
        return int(n_union.value)

    def brick_mask_or(self, dst, src, n=None):
        self._check(self.lib.cpm_brick_mask_or(self.h, self._ptr(dst), self._ptr(src), int(dst.numel() if n is None else n), self._stream()))

    def bricklist_reduce_create(self, comm, grid: "GridDesc", root: int = 0) -> "BricklistReduce":
        h = C.c_void_p()
        self._check(self.lib.cpm_bricklist_reduce_create(self.h, comm.h, C.byref(grid), root, C.byref(h)))
        return BricklistReduce(self, comm, h, root)

    def sparse_reduce_create(self, comm: "Comm", grid: GridDesc) -> "SparseReduce":
        h = C.c_void_p()
        self._check(self.lib.cpm_sparse_reduce_create(self.h, comm.h, C.byref(grid), C.byref(h)))
        return SparseReduce(self, comm, h)

    # -- launch order of the trace
    def trace_order_create(self, n_light_samples: int) -> "TraceOrder":
        return TraceOrder(self, n_light_samples)

    def trace_set_order(self, order, measure: bool = True):
        self._check(self.lib.cpm_trace_set_order(self.h, order.h if order is not None else None, int(bool(measure))))

    # -- OpenGL sharing (needs the host's GL context current on this thread; CpmError(CPM_ERR_UNSUPPORTED) without one)
    def gl_available(self) -> bool:
        return bool(self.lib.cpm_gl_available(self.h))

    def gl_register_buffer(self, gl_buffer: int, read_only: bool = False) -> "GLResource":
        h = C.c_void_p()
        self._check(self.lib.cpm_gl_register_buffer(self.h, int(gl_buffer), int(read_only), C.byref(h)))
        return GLResource(self, h)

    def gl_acquire(self, resources):
        arr = (C.c_void_p * len(resources))(*[r.h for r in resources])
        self._check(self.lib.cpm_gl_acquire(self.h, arr, len(resources), self._stream()))

    def gl_release(self, resources):
        arr = (C.c_void_p * len(resources))(*[r.h for r in resources])
        self._check(self.lib.cpm_gl_release(self.h, arr, len(resources), self._stream()))

    def gl_copy_to_buffer(self, light_volume, buffer: "GLResource", texel: int = CPM_GL_TEXEL_F32):
        self._check(self.lib.cpm_gl_copy_to_buffer(self.h, self._ptr(light_volume), light_volume.numel(), int(texel), buffer.h, self._stream()))

    def light_volume_texels(self, light_volume, out, texel: int = CPM_GL_TEXEL_F32):
        """The light volume as float32 / float16 texels in `out` (a device tensor of light_volume.numel() elements)."""
        self._check(self.lib.cpm_light_volume_texels(self.h, self._ptr(light_volume), light_volume.numel(), int(texel), self._ptr(out), self._stream()))

    # -- temporal interpolation
    def mix_buffers(self, x, y, a, out, kind=None):
        """out = mix(x, y, a); float32 tensors, or (n, 2) uint16 min/max pairs (passed as int16/uint16 tensors)."""
        if kind is None:
            kind = CPM_MIX_F32 if x.dtype == self.torch.float32 else CPM_MIX_U16X2
        n = x.numel() if kind == CPM_MIX_F32 else x.numel() // 2
        self._check(self.lib.cpm_mix_buffers(self.h, self._ptr(x), self._ptr(y), float(a), n, kind, self._ptr(out), self._stream()))

    def volume_mix(self, v0, v1, weight, out):
        self._check(self.lib.cpm_volume_mix(self.h, v0.h, v1.h, float(weight), out.h, self._stream()))

    # -- correlated
    def volume_minmax(self, vol, region, out):
        self._check(self.lib.cpm_volume_minmax(self.h, vol.h, region, self._ptr(out), self._stream()))

    def volume_step(self, cur, nxt, region, diff_out, minmax_out):
        """mean |next - cur| bricks and the min / max bricks of `next` in one pass (cpm_volume_step)."""
        self._check(self.lib.cpm_volume_step(self.h, cur.h, nxt.h, region, self._ptr(diff_out), self._ptr(minmax_out), self._stream()))

    def volume_difference(self, cur, nxt, region, out):
        self._check(self.lib.cpm_volume_difference(self.h, cur.h, nxt.h, region, self._ptr(out), self._stream()))

    def importance_tf(self, minmax, n_cells, positions, colors, out, prev_minmax=None, volume_diff=None, occupancy=None):
        """occupancy (optional int32 tensor of 2 * ceil(n_cells / 64) words): the grid's occupancy bits from the same launch."""
        import numpy as np
        positions = np.ascontiguousarray(positions, dtype=np.float32)
        colors = np.ascontiguousarray(colors, dtype=np.float32)
        self._check(self.lib.cpm_importance_tf_occupancy(self.h, self._ptr(minmax), self._ptr(prev_minmax), self._ptr(volume_diff),
                                                         n_cells, positions.ctypes.data, colors.ctypes.data, positions.shape[0],
                                                         self._ptr(out), self._ptr(occupancy),
                                                         self._stream()))   # (the call has consumed the host arrays on return)

    def photon_importance(self, importance_grid, grid_dims, cell_size, texture_to_index, photons, photon_offset,
                          light_samples, isect, n_light_samples, max_interactions, total_photons, importances,
                          fix_exit_point=False):
        self._check(self.lib.cpm_photon_importance(
            self.h, self._ptr(importance_grid), C.byref((C.c_int32 * 3)(*grid_dims)), C.byref((C.c_float * 3)(*cell_size)),
            C.byref((C.c_float * 16)(*texture_to_index)), self._ptr(photons), photon_offset, self._ptr(light_samples),
            self._ptr(isect), n_light_samples, max_interactions, total_photons, int(fix_exit_point),
            self._ptr(importances), self._stream()))

    def photon_importance_retrace_lights(self, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, aabb, params, spans,
                                         importances, rng_state, photons, old_photons, fix_exit_point=False, tf_scattering=None):
        """photon_importance_retrace for several lights in one launch (spans: Context.light_spans(...))."""
        c = self.ctx
        c._check(c.lib.cpm_photon_importance_retrace_lights(
            c.h, self.h, c._ptr(importance_grid), C.byref((C.c_int32 * 3)(*grid_dims)), C.byref((C.c_float * 3)(*cell_size)),
            C.byref((C.c_float * 16)(*texture_to_index)), vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), C.cast(spans, C.c_void_p), len(spans), int(fix_exit_point),
            c._ptr(importances), c._ptr(rng_state), c._ptr(photons), c._ptr(old_photons), c._stream()))

    def photon_importance_equal(self, photon_offset, n_light_samples, percentage, iteration, importances):
        self._check(self.lib.cpm_photon_importance_equal(self.h, photon_offset, n_light_samples, percentage, iteration,
                                                         self._ptr(importances), self._stream()))

    def reset_importance(self, importances, offset, n):
        self._check(self.lib.cpm_reset_importance(self.h, self._ptr(importances), offset, n, self._stream()))

    def select_changed(self, importances, indices_out, n_changed):
        self._check(self.lib.cpm_select_changed(self.h, self._ptr(importances), importances.numel(), self._ptr(indices_out),
                                                self._ptr(n_changed), self._stream()))

    def select_recompute(self, importances, indices_out, n_changed):
        self._check(self.lib.cpm_select_recompute(self.h, self._ptr(importances), importances.numel(), self._ptr(indices_out),
                                                  self._ptr(n_changed), self._stream()))

    # -- the correlated update without a host round trip
    def selection_create(self, max_photons: int) -> "Selection":
        h = C.c_void_p()
        self._check(self.lib.cpm_selection_create(self.h, max_photons, C.byref(h)))
        return Selection(self, h)

    def trace_selected(self, vol, tf, aabb, params: TraceParams, light_samples, isect, indices, selection: "Selection", max_indices,
                       rng_state, photons, old_photons=None, reset_importances=None, tf_scattering=None):
        self._check(self.lib.cpm_trace_selected(
            self.h, vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), self._ptr(light_samples), self._ptr(isect),
            self._ptr(indices), selection.count_device, max_indices, self._ptr(old_photons), self._ptr(reset_importances),
            self._ptr(rng_state), self._ptr(photons), self._stream()))

    def splat_delta(self, old_photons, old_stride, photons, indices, selection: "Selection", max_indices, grid, radius, scale,
                    n_photons, n_interactions, out, apply_below=0, brick_mask=None):
        self._check(self.lib.cpm_splat_delta(self.h, self._ptr(old_photons), old_stride, self._ptr(photons), self._ptr(indices),
                                             selection.count_device, max_indices, apply_below, C.byref(grid), radius, scale,
                                             n_photons, n_interactions, self._ptr(brick_mask), self._ptr(out), self._stream()))


class Selection:
    """cpm_selection: the changed-photon selection of one correlated update (tile counts, lists, the count's mailbox)."""

    def __init__(self, ctx: "Context", h):
        self.ctx, self.h = ctx, h
        self.count_device = C.c_void_p(ctx.lib.cpm_selection_count_device(h))

    def begin(self):
        self.ctx._check(self.ctx.lib.cpm_selection_begin(self.ctx.h, self.h))

    def set_occupancy(self, importance_grid, occupancy):
        """The grid's occupancy bits from Context.importance_tf(..., occupancy=...): this selection's launches over that grid use them."""
        c = self.ctx
        c._check(c.lib.cpm_selection_set_occupancy(c.h, self.h, c._ptr(importance_grid), c._ptr(occupancy)))

    def photon_importance(self, importance_grid, grid_dims, cell_size, texture_to_index, photons, photon_offset, light_samples, isect,
                          n_light_samples, max_interactions, total_photons, importances, fix_exit_point=False):
        c = self.ctx
        c._check(c.lib.cpm_photon_importance_select(
            c.h, self.h, c._ptr(importance_grid), C.byref((C.c_int32 * 3)(*grid_dims)), C.byref((C.c_float * 3)(*cell_size)),
            C.byref((C.c_float * 16)(*texture_to_index)), c._ptr(photons), photon_offset, c._ptr(light_samples), c._ptr(isect),
            n_light_samples, max_interactions, total_photons, int(fix_exit_point), c._ptr(importances), c._stream()))

    def photon_importance_retrace(self, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, aabb, params, light_samples, isect,
                                  importances, rng_state, photons, old_photons, fix_exit_point=False, tf_scattering=None):
        """Detector + threshold + tracer of one light in one launch (cpm_photon_importance_retrace)."""
        c = self.ctx
        c._check(c.lib.cpm_photon_importance_retrace(
            c.h, self.h, c._ptr(importance_grid), C.byref((C.c_int32 * 3)(*grid_dims)), C.byref((C.c_float * 3)(*cell_size)),
            C.byref((C.c_float * 16)(*texture_to_index)), vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), c._ptr(light_samples), c._ptr(isect), int(fix_exit_point),
            c._ptr(importances), c._ptr(rng_state), c._ptr(photons), c._ptr(old_photons), c._stream()))

    def photon_importance_retrace_lights(self, importance_grid, grid_dims, cell_size, texture_to_index, vol, tf, aabb, params, spans,
                                         importances, rng_state, photons, old_photons, fix_exit_point=False, tf_scattering=None):
        """photon_importance_retrace for several lights in one launch (spans: Context.light_spans(...))."""
        c = self.ctx
        c._check(c.lib.cpm_photon_importance_retrace_lights(
            c.h, self.h, c._ptr(importance_grid), C.byref((C.c_int32 * 3)(*grid_dims)), C.byref((C.c_float * 3)(*cell_size)),
            C.byref((C.c_float * 16)(*texture_to_index)), vol.h, tf.h, tf_scattering.h if tf_scattering is not None else None,
            C.byref((C.c_float * 8)(*aabb)), C.byref(params), C.cast(spans, C.c_void_p), len(spans), int(fix_exit_point),
            c._ptr(importances), c._ptr(rng_state), c._ptr(photons), c._ptr(old_photons), c._stream()))

    def photon_importance_equal(self, photon_offset, n_light_samples, percentage, iteration, importances):
        c = self.ctx
        c._check(c.lib.cpm_photon_importance_equal_select(c.h, self.h, photon_offset, n_light_samples, percentage, iteration,
                                                          c._ptr(importances), c._stream()))

    def finish(self, indices_out):
        c = self.ctx
        c._check(c.lib.cpm_selection_finish(c.h, self.h, c._ptr(indices_out), c._stream()))

    def count(self) -> int:
        """The count of the last finish(), read from its host mailbox (no stream synchronisation)."""
        n = C.c_int32(0)
        self.ctx._check(self.ctx.lib.cpm_selection_count(self.ctx.h, self.h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self.h:
            self.ctx.lib.cpm_selection_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SparseReduce:
    """cpm_sparse_reduce: the sum of the ranks' light volumes over the union of their non-zero 4x4x4 bricks, enqueued without
    a host wait (cpm.h, cpm_allreduce_grid_sparse)."""

    def __init__(self, ctx: "Context", comm: "Comm", h):
        self.ctx, self.comm, self.h = ctx, comm, h

    @property
    def n_bricks(self) -> int:
        return int(self.ctx.lib.cpm_sparse_reduce_bricks(self.h))

    def start(self, partial, total=None, brick_mask=None, root: int = -1, capacity: int = 0, mask_is_nonzero: bool = False) -> int:
        """Enqueue on the current stream; returns the ticket.  brick_mask: the bricks an update touched (default) or, with
        mask_is_nonzero, the non-zero bricks of `partial` as cpm_gather_fast_marked wrote them."""
        ticket = C.c_uint64(0)
        tot = partial if total is None else total
        self.ctx._check(self.ctx.lib.cpm_allreduce_grid_sparse(self.ctx.h, self.h, self.ctx._ptr(partial), self.ctx._ptr(tot),
                                                               self.ctx._ptr(brick_mask) if brick_mask is not None else None,
                                                               1 if mask_is_nonzero else 0, root, capacity, C.byref(ticket), self.ctx._stream()))
        return int(ticket.value)

    def complete(self, ticket: int) -> SparseReduceInfo:
        """Before `total` of the ticket is read (on the current stream): the dense sum after an overflow; the ticket's figures."""
        info = SparseReduceInfo()
        self.ctx._check(self.ctx.lib.cpm_sparse_reduce_complete(self.ctx.h, self.h, ticket, self.ctx._stream(), C.byref(info)))
        return info

    def close(self):
        if self.h:
            self.ctx.lib.cpm_sparse_reduce_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BricklistReduce:
    """cpm_bricklist_reduce: the frame's reduce to the display GPU as per-rank lists of non-zero 4x4x4 bricks (cpm.h,
    cpm_reduce_grid_bricklists): for shards whose bricks are (nearly) disjoint -- contiguous photon ranges."""

    def __init__(self, ctx: "Context", comm: "Comm", h, root: int):
        self.ctx, self.comm, self.h, self.root = ctx, comm, h, root

    @property
    def n_bricks(self) -> int:
        return int(self.ctx.lib.cpm_bricklist_reduce_bricks(self.h))

    def start(self, grid, nonzero_bricks=None) -> int:
        """Enqueue on the current stream (in place: the root's grid becomes the sum, the others' are read); returns the ticket."""
        ticket = C.c_uint64(0)
        self.ctx._check(self.ctx.lib.cpm_reduce_grid_bricklists(self.ctx.h, self.h, self.ctx._ptr(grid),
                                                                self.ctx._ptr(nonzero_bricks) if nonzero_bricks is not None else None,
                                                                C.byref(ticket), self.ctx._stream()))
        return int(ticket.value)

    # -- the same exchange step by step (no dense grid on a sender): open -> fill the segment -> exchange -> complete
    def open(self):
        """A ticket and -- on a rank other than the root -- where its segment lies: (ticket, BricklistSegment); segment.segment is None at
        the root and with one rank (gather into the grid there).  Host only: nothing is enqueued."""
        ticket, seg = C.c_uint64(0), BricklistSegment()
        self.ctx._check(self.ctx.lib.cpm_bricklist_reduce_open(self.ctx.h, self.h, C.byref(ticket), C.byref(seg)))
        return int(ticket.value), seg

    def pack_grid(self, ticket: int, grid, nonzero_bricks=None):
        """Sender: the non-zero 4x4x4 bricks of a dense grid -> the ticket's segment (one launch on the current stream; nothing at the root)."""
        self.ctx._check(self.ctx.lib.cpm_bricklist_pack_grid(self.ctx.h, self.h, ticket, self.ctx._ptr(grid),
                                                             self.ctx._ptr(nonzero_bricks) if nonzero_bricks is not None else None, self.ctx._stream()))

    def exchange(self, ticket: int, root_grid=None):
        """Sender: one send of the segment; root: the receives and the two launches that add them into root_grid (current stream)."""
        self.ctx._check(self.ctx.lib.cpm_bricklist_reduce_exchange(self.ctx.h, self.h, ticket,
                                                                   self.ctx._ptr(root_grid) if root_grid is not None else None, self.ctx._stream()))

    def complete(self, ticket: int) -> BricklistInfo:
        """Before the ticket's grid is read (root) or gathered into again (every rank): the exchange repeated at exact size where a
        rank's list had outgrown its segment; the ticket's figures."""
        info = BricklistInfo()
        self.ctx._check(self.ctx.lib.cpm_bricklist_reduce_complete(self.ctx.h, self.h, ticket, self.ctx._stream(), C.byref(info)))
        return info

    def close(self):
        if self.h:
            self.ctx.lib.cpm_bricklist_reduce_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """One end of the photon-shard communicator (cpm_comm): see cpm.h, multi-GPU."""

    def __init__(self, ctx: "Context", h):
        self.ctx, self.h = ctx, h

    @property
    def rank(self):
        return int(self.ctx.lib.cpm_comm_rank(self.h))

    @property
    def size(self):
        return int(self.ctx.lib.cpm_comm_size(self.h))

    def close(self):
        if self.h:
            self.ctx.lib.cpm_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Volume:
    def __init__(self, ctx: Context, h, desc: VolumeDesc, owned: bool = True):
        """owned=False: a volume something else destroys (a cpm_volume_stream's slot)."""
        self.ctx, self.h, self.desc, self.owned = ctx, h, desc, owned

    @property
    def dims(self):
        return tuple(self.desc.dims)

    def update(self, voxels):
        import numpy as np
        if isinstance(voxels, np.ndarray):
            voxels = np.ascontiguousarray(voxels)
            ptr, is_dev = C.c_void_p(voxels.ctypes.data), 0
        else:
            ptr, is_dev = self.ctx._ptr(voxels), 1
        self.ctx._check(self.ctx.lib.cpm_volume_update(self.ctx.h, self.h, ptr, is_dev, self.ctx._stream()))

    def download(self):
        """Voxels as a numpy array [z, y, x] (blocking device -> host copy)."""
        import numpy as np
        dt = {0: np.uint8, 1: np.uint16, 2: np.float32}[int(self.desc.dtype)]
        out = np.empty(tuple(self.desc.dims)[::-1], dtype=dt)
        self.ctx._check(self.ctx.lib.cpm_volume_download(self.ctx.h, self.h, C.c_void_p(out.ctypes.data), self.ctx._stream()))
        return out

    def __del__(self):
        try:
            if self.h and self.ctx.h and self.owned:
                self.ctx.lib.cpm_volume_destroy(self.ctx.h, self.h)
        except Exception:
            pass


class PinnedSequence:
    """The steps of a time-varying sequence in pinned host memory (cpm_pinned_alloc): what cpm_volume_stream copies from without holding the
    calling thread.  steps[t] is a numpy view [z, y, x] of step t."""

    def __init__(self, ctx: Context, volumes):
        import numpy as np
        self.ctx, self.steps, self._ptrs = ctx, [], []
        for v in volumes:
            v = np.ascontiguousarray(v)
            p = C.c_void_p()
            ctx._check(ctx.lib.cpm_pinned_alloc(ctx.h, v.nbytes, C.byref(p)))
            self._ptrs.append(p)
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(v.nbytes,)).view(v.dtype).reshape(v.shape)
            a[...] = v
            self.steps.append(a)

    def __len__(self):
        return len(self.steps)

    def pointer(self, t: int):
        return self._ptrs[t]

    def close(self):
        self.steps = []
        for p in self._ptrs:
            self.ctx.lib.cpm_pinned_free(self.ctx.h, p)
        self._ptrs = []

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class VolumeStream:
    """cpm_volume_stream: a ring of device volumes over a host-resident sequence, uploads on the library's copy stream (cpm.h)."""

    def __init__(self, ctx: Context, like, n_slots: int = 3):
        """like: a numpy array [z, y, x] of the steps' shape and type (or a VolumeDesc)."""
        import numpy as np
        self.ctx = ctx
        if isinstance(like, VolumeDesc):
            self.desc = like
        else:
            code = {np.dtype(np.uint8): CPM_U8, np.dtype(np.uint16): CPM_U16, np.dtype(np.float32): CPM_F32}[np.dtype(like.dtype)]
            self.desc = default_volume_desc(like.shape[::-1], code)
        self.h = C.c_void_p()
        ctx._check(ctx.lib.cpm_volume_stream_create(ctx.h, C.byref(self.desc), n_slots, C.byref(self.h)))

    @staticmethod
    def _host(voxels):
        if voxels is None:
            return None
        if isinstance(voxels, C.c_void_p):
            return voxels
        return C.c_void_p(voxels.ctypes.data)

    def prefetch(self, tag: int, voxels):
        """voxels: a numpy array (pinned: PinnedSequence.steps[t]) or a raw pointer; the consumer is the current stream."""
        self.ctx._check(self.ctx.lib.cpm_volume_stream_prefetch(self.ctx.h, self.h, tag, self._host(voxels), self.ctx._stream()))

    def acquire(self, tag: int, voxels=None) -> "Volume":
        h = C.c_void_p()
        self.ctx._check(self.ctx.lib.cpm_volume_stream_acquire(self.ctx.h, self.h, tag, self._host(voxels), self.ctx._stream(), C.byref(h)))
        return Volume(self.ctx, h, self.desc, owned=False)

    def stats(self) -> VolumeStreamInfo:
        info = VolumeStreamInfo()
        self.ctx._check(self.ctx.lib.cpm_volume_stream_stats(self.ctx.h, self.h, C.byref(info)))
        return info

    def close(self):
        if self.h:
            self.ctx.lib.cpm_volume_stream_destroy(self.ctx.h, self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.close()
        except Exception:
            pass


class TransferFunction:
    def __init__(self, ctx: Context, h, width):
        self.ctx, self.h, self.width = ctx, h, width

    def update(self, rgba):
        import numpy as np
        if isinstance(rgba, np.ndarray):
            rgba = np.ascontiguousarray(rgba, dtype=np.float32)
            ptr, is_dev = C.c_void_p(rgba.ctypes.data), 0
        else:
            ptr, is_dev = self.ctx._ptr(rgba), 1
        self.ctx._check(self.ctx.lib.cpm_tf_update(self.ctx.h, self.h, ptr, is_dev, self.ctx._stream()))

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.cpm_tf_destroy(self.ctx.h, self.h)
        except Exception:
            pass
