"""ctypes binding of oracle/libcpm_oracle.so (the CPU restatement) and, when present,
oracle/_ref/libcpm_ref.so (the reference's own OpenCL C compiled for x86-64).

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
ORACLE_DIR = REPO / "oracle"
ORACLE_LIB = ORACLE_DIR / "libcpm_oracle.so"
REF_LIB = ORACLE_DIR / "_ref" / "libcpm_ref.so"


class OVolume(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3), ("dtype", C.c_int32), ("format_scaling", C.c_float),
                ("format_offset", C.c_float), ("texture_to_index", C.c_float * 16),
                ("index_to_texture", C.c_float * 16), ("voxels", C.c_void_p)]


class OTraceParams(C.Structure):
    _fields_ = [("material", C.c_float * 4), ("step_size", C.c_float), ("photon_offset", C.c_int32),
                ("n_light_samples", C.c_int32), ("max_interactions", C.c_int32),
                ("total_photons", C.c_int32), ("shading_type", C.c_int32), ("flags", C.c_int32),
                ("iteration", C.c_int32), ("batch", C.c_int32)]


class OGrid(C.Structure):
    _fields_ = [("dims", C.c_int32 * 3), ("channels", C.c_int32),
                ("texture_to_index", C.c_float * 16), ("index_to_texture", C.c_float * 16)]


def build_oracle():
    subprocess.run(["make", "-C", str(ORACLE_DIR), "-s"], check=True)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_matrices(dims):
    """Same float32 construction as cpm_volume_desc_default / cpm_grid_desc_default."""
    t2i = np.zeros(16, np.float32)
    i2t = np.zeros(16, np.float32)
    for a in range(3):
        t2i[5 * a] = np.float32(dims[a])
        t2i[12 + a] = np.float32(-0.5)
        i2t[5 * a] = np.float32(1.0) / np.float32(dims[a])
        i2t[12 + a] = np.float32(0.5) / np.float32(dims[a])
    t2i[15] = i2t[15] = 1.0
    return t2i, i2t


class Oracle:
    def __init__(self):
        if not ORACLE_LIB.exists():
            build_oracle()
        self.lib = C.CDLL(str(ORACLE_LIB))
        L = self.lib
        L.cpmo_log.restype = C.c_float
        L.cpmo_log.argtypes = [C.c_float]
        L.cpmo_acos.restype = C.c_float
        L.cpmo_acos.argtypes = [C.c_float]
        L.cpmo_atan2.restype = C.c_float
        L.cpmo_atan2.argtypes = [C.c_float, C.c_float]
        L.cpmo_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.cpmo_density_kernel.restype = C.c_float
        L.cpmo_density_kernel.argtypes = [C.c_float]
        L.cpmo_relative_irradiance_scale.restype = C.c_float
        L.cpmo_relative_irradiance_scale.argtypes = [C.c_double, C.c_double]
        L.cpmo_sample_volume.restype = C.c_float
        L.cpmo_sample_volume.argtypes = [C.POINTER(OVolume), C.c_float, C.c_float, C.c_float]
        L.cpmo_sample_tf_alpha.restype = C.c_float
        L.cpmo_sample_tf_alpha.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.cpmo_seed_streams.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.cpmo_random_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.cpmo_glibc_rand_sequence.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
        L.cpmo_sort_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.cpmo_sort_keys.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        L.cpmo_select_recompute.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.cpmo_set_threads.argtypes = [C.c_int]
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        L.cpmo_convex_hull_2d.restype, L.cpmo_convex_hull_2d.argtypes = C.c_int, [fp, C.c_int, fp]
        L.cpmo_minimum_bounding_rectangle.restype, L.cpmo_minimum_bounding_rectangle.argtypes = None, [fp, C.c_int, fp]
        L.cpmo_fit_obb.restype, L.cpmo_fit_obb.argtypes = None, [fp, C.c_int, fp, fp, fp]
        L.cpmo_tf_difference_points.restype = C.c_int
        L.cpmo_tf_difference_points.argtypes = [dp, fp, C.c_int, dp, fp, C.c_int, C.c_float, C.c_int, fp, fp]

    # ---- helpers
    def set_threads(self, n):
        self.lib.cpmo_set_threads(int(n))

    def volume(self, voxels: np.ndarray, format_scaling=0.0, format_offset=0.0) -> OVolume:
        code = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}[voxels.dtype]
        v = OVolume()
        dims = voxels.shape[::-1]
        for a in range(3):
            v.dims[a] = dims[a]
        v.dtype = code
        v.format_scaling = format_scaling
        v.format_offset = format_offset
        t2i, i2t = default_matrices(dims)
        v.texture_to_index[:] = t2i.tolist()
        v.index_to_texture[:] = i2t.tolist()
        v._keep = np.ascontiguousarray(voxels)
        v.voxels = v._keep.ctypes.data
        return v

    def grid(self, dims, channels=1) -> OGrid:
        g = OGrid()
        for a in range(3):
            g.dims[a] = dims[a]
        g.channels = channels
        t2i, i2t = default_matrices(dims)
        g.texture_to_index[:] = t2i.tolist()
        g.index_to_texture[:] = i2t.tolist()
        return g

    # ---- math
    def log(self, x):
        return np.array([self.lib.cpmo_log(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def sincos(self, x):
        s, c = C.c_float(), C.c_float()
        out = []
        for v in np.asarray(x, np.float32).ravel():
            self.lib.cpmo_sincos(float(v), C.byref(s), C.byref(c))
            out.append((s.value, c.value))
        return np.array(out, np.float32)

    def encode_direction(self, d):
        d = np.ascontiguousarray(d, np.float32)
        out = np.zeros(2, np.float32)
        self.lib.cpmo_encode_direction(_p(d), _p(out))
        return out

    def decode_direction(self, a):
        a = np.ascontiguousarray(a, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.cpmo_decode_direction(_p(a), _p(out))
        return out

    # ---- RNG
    def glibc_rand_sequence(self, seed, n):
        out = np.zeros(n, np.uint32)
        self.lib.cpmo_glibc_rand_sequence(seed, _p(out), n)
        return out

    def seed_streams(self, state: np.ndarray, gap=1 << 40):
        assert state.dtype == np.uint32 and state.flags.c_contiguous
        self.lib.cpmo_seed_streams(_p(state), state.shape[0], gap)

    def random_fill(self, state: np.ndarray, draws: int):
        out = np.zeros((draws, state.shape[0]), np.float32)
        self.lib.cpmo_random_fill(_p(state), state.shape[0], draws, _p(out))
        return out

    # ---- emission
    def uniform_samples_2d(self, nx, ny):
        out = np.zeros((nx * ny, 4), np.float32)
        self.lib.cpmo_uniform_samples_2d(nx, ny, _p(out))
        return out

    @staticmethod
    def _f4(v):
        v = list(np.asarray(v, np.float32).ravel()) + [0.0] * 4
        return (C.c_float * 4)(*v[:4])

    def directional_light_samples(self, samples, radiance, direction, origin, u, v, area):
        n = samples.shape[0]
        out = np.zeros((n, 8), np.float32)
        self.lib.cpmo_directional_light_samples(_p(samples), n, self._f4(radiance), self._f4(direction), self._f4(origin),
                                                self._f4(u), self._f4(v), C.c_float(area), _p(out))
        return out

    def point_light_samples(self, samples, radiance, position):
        n = samples.shape[0]
        out = np.zeros((n, 8), np.float32)
        self.lib.cpmo_point_light_samples(_p(samples), n, self._f4(radiance), self._f4(position), _p(out))
        return out

    def light_sample_box_intersection(self, ls, aabb):
        n = ls.shape[0]
        out = np.zeros((n, 2), np.float32)
        self.lib.cpmo_light_sample_box_intersection(_p(ls), n, (C.c_float * 8)(*aabb), _p(out))
        return out

    def light_sample_mesh_intersection(self, vertices, indices, ls):
        n = ls.shape[0]
        out = np.zeros((n, 2), np.float32)
        vertices = np.ascontiguousarray(vertices, np.float32)
        indices = np.ascontiguousarray(indices, np.int32)
        self.lib.cpmo_light_sample_mesh_intersection(_p(vertices), _p(indices), indices.size, _p(ls), n, _p(out))
        return out

    # ---- trace
    def trace(self, vol: OVolume, tf_rgba, aabb, params: OTraceParams, ls, isect, rng, photons,
              recompute_indices=None, n_recompute=0, tf_scattering=None):
        steps = C.c_uint64(0)
        tf_rgba = np.ascontiguousarray(tf_rgba, np.float32)
        self.lib.cpmo_trace(C.byref(vol), _p(tf_rgba), tf_rgba.shape[0], _p(tf_scattering), (C.c_float * 8)(*aabb),
                            C.byref(params), _p(ls), _p(isect), _p(recompute_indices), n_recompute, _p(rng), _p(photons),
                            C.byref(steps))
        return steps.value

    # ---- light volume
    def relative_irradiance_scale(self, r, n):
        return float(self.lib.cpmo_relative_irradiance_scale(r, n))

    def splat(self, photons, total, grid: OGrid, radius, scale, out):
        self.lib.cpmo_splat(_p(photons), total, C.byref(grid), C.c_float(radius), C.c_float(scale), _p(out))

    def splat_selected(self, photons, indices, grid, radius, scale, multiplier, n_photons, n_interactions, out):
        self.lib.cpmo_splat_selected(_p(photons), _p(indices), indices.size, C.byref(grid), C.c_float(radius), C.c_float(scale),
                                     C.c_float(multiplier), n_photons, n_interactions, _p(out))

    def copy_indexed_photons(self, photons, indices, multiplier, n_photons, n_interactions, aligned, out_offset=0):
        self.lib.cpmo_copy_indexed_photons(_p(photons), _p(indices), indices.size, C.c_float(multiplier), n_photons,
                                           n_interactions, _p(aligned), out_offset)

    def sort_pairs(self, keys, values, key_bits=0):
        self.lib.cpmo_sort_pairs(_p(keys), _p(values), keys.size, key_bits)

    def sort_keys(self, keys, key_bits=0):
        self.lib.cpmo_sort_keys(_p(keys), keys.size, key_bits)

    def bin(self, photons, n, grid: OGrid):
        cells = grid.dims[0] * grid.dims[1] * grid.dims[2]
        order = np.zeros(n, np.uint32)
        cell_start = np.zeros(cells + 1, np.uint32)
        sorted_pp = np.zeros((n, 4 if grid.channels == 1 else 8), np.float32)
        self.lib.cpmo_bin(_p(photons), n, C.byref(grid), _p(order), _p(cell_start), _p(sorted_pp))
        return order, cell_start, sorted_pp

    def gather(self, sorted_pp, cell_start, n, grid: OGrid, radius, scale, out, accumulate=False):
        self.lib.cpmo_gather(_p(sorted_pp), _p(cell_start), n, C.byref(grid), C.c_float(radius), C.c_float(scale),
                             int(accumulate), _p(out))

    def gather_fast(self, photons, n, grid: OGrid, radius, scale, out, accumulate=False):
        """The tolerance-mode formulation (cpm_bin_fast + cpm_gather_fast) restated: fixed-point sums, order-free."""
        self.lib.cpmo_gather_fast(_p(photons), n, C.byref(grid), C.c_float(radius), C.c_float(scale), int(accumulate), _p(out))

    # ---- temporal interpolation
    def mix_f32(self, x, y, a):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        out = np.empty_like(x)
        self.lib.cpmo_mix_f32(_p(x), _p(y), C.c_float(a), C.c_size_t(x.size), _p(out))
        return out

    def mix_u16x2(self, x, y, a):
        x = np.ascontiguousarray(x, np.uint16); y = np.ascontiguousarray(y, np.uint16)
        out = np.empty_like(x)
        self.lib.cpmo_mix_u16x2(_p(x), _p(y), C.c_float(a), C.c_size_t(x.size // 2), _p(out))
        return out

    def volume_mix(self, a: OVolume, b: OVolume, weight, like):
        out = np.empty_like(like)
        self.lib.cpmo_volume_mix(C.byref(a), C.byref(b), C.c_float(weight), _p(out))
        return out

    # ---- correlated
    def volume_minmax(self, vol: OVolume, region):
        o = [(vol.dims[a] + region - 1) // region for a in range(3)]
        out = np.zeros((o[0] * o[1] * o[2], 2), np.uint16)
        self.lib.cpmo_volume_minmax(C.byref(vol), region, _p(out))
        return out

    def volume_difference(self, a: OVolume, b: OVolume, region):
        o = [(a.dims[k] + region - 1) // region for k in range(3)]
        out = np.zeros(o[0] * o[1] * o[2], np.float32)
        self.lib.cpmo_volume_difference(C.byref(a), C.byref(b), region, _p(out))
        return out

    def importance_tf(self, minmax, positions, colors, prev=None, diff=None):
        n = minmax.shape[0]
        out = np.zeros(n, np.float32)
        positions = np.ascontiguousarray(positions, np.float32)
        colors = np.ascontiguousarray(colors, np.float32)
        self.lib.cpmo_importance_tf(_p(minmax), _p(prev), _p(diff), n, _p(positions), _p(colors), positions.size, _p(out))
        return out

    def photon_importance(self, grid, grid_dims, cell_size, t2i, photons, photon_offset, ls, isect, n_ls, max_inter,
                          total, importances, fix_exit_point=False):
        self.lib.cpmo_photon_importance(_p(grid), (C.c_int32 * 3)(*grid_dims), (C.c_float * 3)(*cell_size),
                                        (C.c_float * 16)(*t2i), _p(photons), photon_offset, _p(ls), _p(isect), n_ls,
                                        max_inter, total, int(fix_exit_point), _p(importances))

    def photon_importance_equal(self, photon_offset, n_ls, percentage, iteration, importances):
        self.lib.cpmo_photon_importance_equal(photon_offset, n_ls, percentage, iteration, _p(importances))

    def select_recompute(self, importances):
        idx = np.zeros(importances.size, np.uint32)
        cnt = C.c_int32(0)
        self.lib.cpmo_select_recompute(_p(importances), importances.size, _p(idx), C.byref(cnt))
        return idx, cnt.value


    def select_changed(self, importances):
        idx = np.zeros(importances.size, np.uint32)
        cnt = C.c_int32(0)
        self.lib.cpmo_select_changed(_p(np.ascontiguousarray(importances, np.uint32)), importances.size, _p(idx), C.byref(cnt))
        return idx, cnt.value


    # ---- host arithmetic (cpm_oracle_host.c)
    @staticmethod
    def _fp(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    def convex_hull_2d(self, points):
        pts = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 2))
        out = np.zeros((len(pts) + 2, 2), np.float32)
        n = self.lib.cpmo_convex_hull_2d(self._fp(pts), len(pts), self._fp(out))
        return out[:n].copy()

    def minimum_bounding_rectangle(self, hull):
        pts = np.ascontiguousarray(np.asarray(hull, np.float32).reshape(-1, 2))
        out = np.zeros(6, np.float32)
        self.lib.cpmo_minimum_bounding_rectangle(self._fp(pts), len(pts), self._fp(out))
        return out

    def fit_obb(self, points, plane_point, unit_normal):
        pts = np.ascontiguousarray(np.asarray(points, np.float32).reshape(-1, 3))
        pp, nn = np.ascontiguousarray(plane_point, np.float32), np.ascontiguousarray(unit_normal, np.float32)
        out = np.zeros(9, np.float32)
        self.lib.cpmo_fit_obb(self._fp(pts), len(pts), self._fp(pp), self._fp(nn), self._fp(out))
        return out[0:3].copy(), out[3:6].copy(), out[6:9].copy()

    def tf_difference_points(self, tf_points, prev_points, eps=1e-4, associated=False):
        def split(points):
            pts = sorted(points, key=lambda q: float(q[0]))
            return (np.ascontiguousarray([float(q[0]) for q in pts], np.float64),
                    np.ascontiguousarray([[np.float32(c) for c in q[1:5]] for q in pts], np.float32).reshape(-1, 4))
        (pa, ca), (pb, cb) = split(tf_points), split(prev_points)
        cap = len(pa) + len(pb) + 2
        pos, col = np.zeros(cap, np.float32), np.zeros((cap, 4), np.float32)
        dp = C.POINTER(C.c_double)
        n = self.lib.cpmo_tf_difference_points(pa.ctypes.data_as(dp), self._fp(ca), len(pa), pb.ctypes.data_as(dp), self._fp(cb), len(pb),
                                               float(eps), int(bool(associated)), self._fp(pos), self._fp(col))
        return (None, None) if n < 0 else (pos[:n].copy(), col[:n].copy())


class Ref:
    """The reference's own kernels (oracle/_ref), when that build exists."""

    def __init__(self):
        if not REF_LIB.exists():
            raise FileNotFoundError(REF_LIB)
        self.lib = C.CDLL(str(REF_LIB))
        lib2 = REF_LIB.with_name("libcpm_ref2.so")  # photon.cl, randomnumbergenerator.cl (oracle/ref_harness2.c)
        self.lib2 = C.CDLL(str(lib2)) if lib2.exists() else None
        self.lib.ref_density_kernel.restype = C.c_float
        self.lib.ref_density_kernel.argtypes = [C.c_float]
        self.lib.ref_generate_per_stream_random_state.argtypes = [C.c_void_p, C.c_uint64, C.c_int]

    def generate_random_state(self, state):
        self.lib.ref_generate_random_state(_p(state), state.shape[0])

    def generate_per_stream_random_state(self, state, gap):
        self.lib.ref_generate_per_stream_random_state(_p(state), gap, state.shape[0])

    def random_fill(self, state, draws):
        out = np.zeros((draws, state.shape[0]), np.float32)
        outu = np.zeros((draws, state.shape[0]), np.uint32)
        self.lib.ref_random_fill(_p(state), state.shape[0], draws, _p(out), _p(outu))
        return out, outu

    def photon_write_read(self, photons8, ids, capacity):
        """writePhoton(photons8[i], buffer, ids[i]) for every i, then readPhoton back: (buffer as written, read-back)."""
        photons8 = np.ascontiguousarray(photons8, np.float32)
        ids = np.ascontiguousarray(ids, np.int32)
        buf = np.full((capacity, 8), np.float32(-7), np.float32)
        out = np.zeros_like(photons8)
        self.lib2.ref_photon_write_read(_p(photons8), _p(ids), ids.size, _p(buf), _p(out))
        return buf, out

    def random_number_kernel(self, state):
        """randomNumberGeneratorKernel: one random_01 per stream, the state loaded from and saved back to the uint2 buffer."""
        out = np.zeros(state.shape[0], np.float32)
        self.lib2.ref_random_number_kernel(_p(state), state.shape[0], _p(out))
        return out

    def density_kernel(self, x):
        return np.array([self.lib.ref_density_kernel(float(v)) for v in np.asarray(x, np.float32).ravel()], np.float32)

    def threshold(self, data, threshold):
        out = np.zeros_like(data)
        self.lib.ref_threshold(_p(data), C.c_uint32(threshold), data.size, _p(out))
        return out

    def index_to_buffer(self, n):
        out = np.zeros(n, np.uint32)
        self.lib.ref_index_to_buffer(_p(out), n)
        return out

    def mix_f32(self, x, y, a, wg=128):
        x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
        out = np.full_like(x, np.float32(-1))
        self.lib.ref_mix_f32(_p(x), _p(y), C.c_float(a), C.c_uint(x.size), _p(out), C.c_uint(wg))
        return out

    def mix_u16x2(self, x, y, a, wg=128):
        x = np.ascontiguousarray(x, np.uint16); y = np.ascontiguousarray(y, np.uint16)
        out = np.full_like(x, 0xffff)
        self.lib.ref_mix_u16x2(_p(x), _p(y), C.c_float(a), C.c_uint(x.size // 2), _p(out), C.c_uint(wg))
        return out
