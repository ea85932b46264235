"""ctypes access to libcpm_host.so's C facade (host/cpm_host_c.cpp): the C++ Processor/Port layer as bench.py drives it.

Load it AFTER torch has initialised its HIP runtime (a second libamdhip64 brought up first does not see the GPU)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import binding as B
from . import build as _build


def load():
    _build.build_host_library(verbose=False)
    lib = C.CDLL(str(B.LIB_PATH.parent / "libcpm_host.so"))
    vp, i32 = C.c_void_p, C.c_int
    sigs = {
        "cpmh_create": (vp, [vp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), vp, i32, i32, i32, i32]),
        "cpmh_destroy": (None, [vp]),
        "cpmh_evaluate": (i32, [vp, i32]),
        "cpmh_set_transfer_function": (None, [vp, vp, i32]),
        "cpmh_set_property_float": (i32, [vp, C.c_char_p, C.c_char_p, C.c_float]),
        "cpmh_set_property_string": (i32, [vp, C.c_char_p, C.c_char_p, C.c_char_p]),
        "cpmh_n_photons": (i32, [vp]),
        "cpmh_n_recomputed": (i32, [vp]),
        "cpmh_last_light_volume_path": (C.c_char_p, [vp]),
        "cpmh_last_tracer_decision": (C.c_char_p, [vp]),
        "cpmh_path_costs": (None, [vp, vp]),
        "cpmh_bench_tf_edits": (i32, [vp, vp, i32, vp, i32, i32, vp, vp]),
        "cpmh_bench_tf_edits_timeline": (i32, [vp, vp, i32, vp, i32, i32, vp]),
        "cpmh_bench_full_frames": (i32, [vp, i32, vp]),
        "cpmh_bench_frames_back_to_back": (i32, [vp, i32, vp, vp]),
        "cpmh_profile_full_frames": (C.c_char_p, [vp, i32]),
        "cpmh_add_light": (i32, [vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
        "cpmh_n_lights": (i32, [vp]),
        "cpmh_set_clip": (None, [vp, i32, i32, i32, i32, i32, i32]),
        "cpmh_sequence_create": (vp, [vp, i32, i32, i32, i32, i32, i32]),
        "cpmh_sequence_destroy": (None, [vp]),
        "cpmh_sequence_keep_on_device": (None, [vp, i32]),
        "cpmh_sequence_stream_stats": (i32, [vp, vp]),
        "cpmh_attach_sequence": (i32, [vp, vp]),
        "cpmh_sequence_step": (i32, [vp, vp, C.c_float, vp]),
        "cpmh_sequence_step_total": (i32, [vp, vp, C.c_float, vp]),
    }
    for name, (res, args) in sigs.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    return lib


class HostNetwork:
    """The CorrelatedPhotonMappingSingleVolume network in C++ (sample generator -> light sampler -> tracer -> light volume,
    min/max -> importance -> tracer): cpmh_create's wiring."""

    def __init__(self, lib, volume_u8, n_side, light_position, light_direction, tf_points, size_option=2, max_scattering=1,
                 correlated=True):
        self.lib = lib
        vol = np.ascontiguousarray(volume_u8)
        pts = np.ascontiguousarray(np.asarray(tf_points, np.float32))
        self.h = lib.cpmh_create(vol.ctypes.data, 0, vol.shape[2], vol.shape[1], vol.shape[0], n_side, n_side,
                                 C.byref((C.c_float * 3)(*light_position)), C.byref((C.c_float * 3)(*light_direction)),
                                 pts.ctypes.data, pts.shape[0], size_option, max_scattering, int(correlated))
        if not self.h:
            raise RuntimeError("cpmh_create failed")

    def evaluate(self, first=False):
        if self.lib.cpmh_evaluate(self.h, int(first)) != 0:
            raise RuntimeError("cpmh_evaluate failed")

    def set_float(self, processor: str, prop: str, value: float):
        if self.lib.cpmh_set_property_float(self.h, processor.encode(), prop.encode(), float(value)) != 0:
            raise KeyError(f"{processor}.{prop}")

    def set_string(self, processor: str, prop: str, value: str):
        if self.lib.cpmh_set_property_string(self.h, processor.encode(), prop.encode(), value.encode()) != 0:
            raise KeyError(f"{processor}.{prop}")

    @property
    def n_photons(self):
        return int(self.lib.cpmh_n_photons(self.h))

    @property
    def last_path(self):
        return self.lib.cpmh_last_light_volume_path(self.h).decode()

    @property
    def last_decision(self):
        return self.lib.cpmh_last_tracer_decision(self.h).decode()

    def path_costs(self):
        c = (C.c_float * 4)()
        self.lib.cpmh_path_costs(self.h, C.byref(c))
        return {"full_trace_ms": c[0], "full_light_volume_ms": c[1], "branch_trace_ms": c[2], "branch_light_volume_ms": c[3]}

    def bench_tf_edits(self, points_a, points_b, reps):
        a = np.ascontiguousarray(np.asarray(points_a, np.float32))
        b = np.ascontiguousarray(np.asarray(points_b, np.float32))
        ms, n = (C.c_double * reps)(), (C.c_int * reps)()
        if self.lib.cpmh_bench_tf_edits(self.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], reps, C.byref(ms), C.byref(n)) != 0:
            raise RuntimeError("cpmh_bench_tf_edits failed")
        return np.array(list(ms)), np.array(list(n))

    def bench_tf_edits_timeline(self, points_a, points_b, reps):
        """Host clock since the edit after: property set, importance, tracer, light volume returned, device idle -> (reps, 5) ms."""
        a = np.ascontiguousarray(np.asarray(points_a, np.float32))
        b = np.ascontiguousarray(np.asarray(points_b, np.float32))
        ms = (C.c_double * (5 * reps))()
        if self.lib.cpmh_bench_tf_edits_timeline(self.h, a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], reps, C.byref(ms)) != 0:
            raise RuntimeError("cpmh_bench_tf_edits_timeline failed")
        return np.array(list(ms)).reshape(reps, 5)

    def bench_full_frames(self, reps):
        """Latency: every frame timed from an idle device until it is idle again."""
        ms = (C.c_double * reps)()
        if self.lib.cpmh_bench_full_frames(self.h, reps, C.byref(ms)) != 0:
            raise RuntimeError("cpmh_bench_full_frames failed")
        return np.array(list(ms))

    def bench_frames_back_to_back(self, reps):
        """Throughput: `reps` full frames enqueued back to back, one synchronisation; (ms per frame, host enqueue ms per frame)."""
        total, host = C.c_double(0), C.c_double(0)
        if self.lib.cpmh_bench_frames_back_to_back(self.h, reps, C.byref(total), C.byref(host)) != 0:
            raise RuntimeError("cpmh_bench_frames_back_to_back failed")
        return total.value / reps, host.value / reps

    def profile_full_frames(self, reps):
        """{kernel name: ms per frame} over `reps` full frames (HIP events around every launch)."""
        text = self.lib.cpmh_profile_full_frames(self.h, reps).decode()
        return {k: float(v) for k, v in (item.split("=") for item in text.split(";") if item)}

    def add_light(self, light_position, light_direction):
        return int(self.lib.cpmh_add_light(self.h, C.byref((C.c_float * 3)(*light_position)), C.byref((C.c_float * 3)(*light_direction))))

    def set_clip(self, x0, x1, y0, y1, z0, z1):
        self.lib.cpmh_set_clip(self.h, x0, x1, y0, y1, z0, z1)

    def close(self):
        if self.h:
            self.lib.cpmh_destroy(self.h)
            self.h = None


class HostSequence:
    def __init__(self, lib, volumes_u8, region=8):
        self.lib = lib
        v = np.ascontiguousarray(volumes_u8)
        self.h = lib.cpmh_sequence_create(v.ctypes.data, 0, v.shape[3], v.shape[2], v.shape[1], v.shape[0], region)
        if not self.h:
            raise RuntimeError("cpmh_sequence_create failed")

    def attach(self, net: HostNetwork):
        if self.lib.cpmh_attach_sequence(net.h, self.h) != 0:
            raise RuntimeError("cpmh_attach_sequence failed")

    def step(self, net: HostNetwork, time: float):
        t = (C.c_double * 2)()
        n = self.lib.cpmh_sequence_step(net.h, self.h, float(time), C.byref(t))
        if n < -1:
            raise RuntimeError("cpmh_sequence_step failed")
        return n, t[0], t[1]

    def step_total(self, net: HostNetwork, time: float):
        """One displayed time, everything enqueued back to back, one synchronisation: (photons re-traced, ms)."""
        t = C.c_double(0)
        n = self.lib.cpmh_sequence_step_total(net.h, self.h, float(time), C.byref(t))
        if n < -1:
            raise RuntimeError("cpmh_sequence_step_total failed")
        return n, t.value

    def keep_on_device(self, keep: bool):
        """False: the elements stay in host memory and are streamed through the player's ring of device volumes (cpm_volume_stream)."""
        self.lib.cpmh_sequence_keep_on_device(self.h, int(bool(keep)))

    def stream_stats(self):
        """{"uploads", "upload_ms", "bytes_per_step", "uploads_at_acquire"} of the streaming player, or None when it keeps the sequence resident."""
        out = (C.c_double * 4)()
        if self.lib.cpmh_sequence_stream_stats(self.h, out) != 0:
            return None
        return {"uploads": int(out[0]), "upload_ms": float(out[1]), "bytes_per_step": int(out[2]), "uploads_at_acquire": int(out[3])}

    def close(self):
        if self.h:
            self.lib.cpmh_sequence_destroy(self.h)
            self.h = None
