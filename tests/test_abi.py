"""The C-ABI library loads without a GPU and exports every symbol include/cpm/*.h declares."""
import ctypes as C
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def declared_functions(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(cpm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(cpm):
    """Both headers: cpm.h, the core (one entry point per call site of the reference's host code), and cpm_ext.h (what this build adds);
    the binding's two lists are the headers' declarations, nothing is declared twice, and the core stays the size of the reference's own
    surface."""
    lib = cpm.binding.load_library()
    core = declared_functions(REPO / "include" / "cpm" / "cpm.h")
    ext = declared_functions(REPO / "include" / "cpm" / "cpm_ext.h")
    assert 30 <= len(core) <= 56 and len(ext) >= 40 and not set(core) & set(ext)
    missing = [f for f in core + ext if not hasattr(lib, f)]
    assert not missing, f"declared in cpm.h / cpm_ext.h but not exported: {missing}"
    assert sorted(cpm.binding.CORE_SYMBOLS) == core, "binding.CORE_SYMBOLS is out of sync with cpm.h"
    assert sorted(cpm.binding.EXT_SYMBOLS) == ext, "binding.EXT_SYMBOLS is out of sync with cpm_ext.h"
    # the experiments' residue is not in the core: no layout modes, no fast formulation, no fused update, no exchanges beyond the dense reduce
    for name in core:
        assert not any(t in name for t in ("_fast", "_layout", "selection", "bricklist", "sparse", "_gl_", "trace_order", "volume_stream", "records_")), name
    for f in declared_functions(REPO / "include" / "cpm" / "cpm_profile.h"):
        assert hasattr(lib, f), f
    assert lib.cpm_abi_version() == 2


def test_struct_layouts_match_header(cpm):
    B = cpm.binding
    assert C.sizeof(B.VolumeDesc) == 4 * (3 + 1 + 2 + 16 + 16)
    assert C.sizeof(B.GridDesc) == 4 * (3 + 1 + 16 + 16)
    assert C.sizeof(B.TraceParams) == 4 * (4 + 1 + 8)


def test_default_descriptors_and_host_helpers(cpm, oracle):
    import numpy as np
    from oracle_binding import default_matrices
    B = cpm.binding
    for dims in ((128, 128, 128), (50, 33, 19), (512, 512, 96)):
        g = B.default_grid_desc(dims, 1)
        v = B.default_volume_desc(dims, B.CPM_U8)
        t2i, i2t = default_matrices(dims)
        assert list(g.texture_to_index) == t2i.tolist() == list(v.texture_to_index)
        assert list(g.index_to_texture) == i2t.tolist() == list(v.index_to_texture)
    assert np.array_equal(B.glibc_rand_sequence(0, 1000), oracle.glibc_rand_sequence(0, 1000))
    for r, n in ((0.0067658, 1048576), (0.02, 65536), (0.5, 7)):
        assert B.relative_irradiance_scale(r, n) == oracle.relative_irradiance_scale(r, n)
    # SURVEY appendix B walk-through: 0.2340 at config 2
    assert abs(B.relative_irradiance_scale(3 ** 0.5 / 256, 1048576) - 0.2340) < 2e-4


def test_sparse_reduce_capacity_policy_is_the_same_on_both_sides(cpm):
    """sharding.sparse_capacity (the CPU mirror of the sparse reduce) and cpm_sparse_reduce_capacity_for (what the library
    sizes its payload with) are one function: pure host arithmetic, no device."""
    import importlib
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    lib = cpm.binding.load_library()
    for nb in (0, 1, 27, 512, 4096, 32768, 262144):
        for prev in (-1, 0, 1, 63, 64, 100, nb // 8, nb // 4, nb // 3, nb // 2, nb):
            assert lib.cpm_sparse_reduce_capacity_for(nb, prev) == sh.sparse_capacity(nb, prev), (nb, prev)
    assert sh.sparse_capacity(32768, 4000) == 5120 and sh.sparse_capacity(32768, -1) == 8192
    assert sh.sparse_capacity(32768, 14000) == 32768  # beyond half of the bricks: dense


def test_bricklist_policy_and_exchange_model_are_host_arithmetic(cpm):
    """sharding.bricklist_capacity / bricklist_segment_bytes mirror cpm_bricklist_capacity_for / cpm_bricklist_segment_bytes (a sender and
    the root must derive the same segment size on their own); sharding.exchange_model prices the three exchanges from brick counts."""
    import importlib
    sh = importlib.import_module(cpm.__name__ + ".sharding")
    lib = cpm.binding.load_library()
    for nb in (0, 1, 27, 60, 512, 4096, 32768, 262144):
        for prev in (-1, 0, 1, 63, 64, 100, nb // 8, nb // 4, nb // 2, nb, 2 * nb):
            assert lib.cpm_bricklist_capacity_for(nb, prev) == sh.bricklist_capacity(nb, prev), (nb, prev)
    for cap in (0, 64, 1024, 7680):
        for ch in (1, 4):
            assert lib.cpm_bricklist_segment_bytes(cap, ch) == sh.bricklist_segment_bytes(cap, ch) == 16 + cap * (256 * ch + 16)
    assert sh.bricklist_capacity(32768, 6077) == 7680 and sh.bricklist_capacity(32768, -1) == 8192 and sh.bricklist_capacity(60, 1000) == 64
    # config 4 at 8 ranks, the counts measured on one GPU (profiles/r05_shard_exchange_bytes_config4.json): slab shards + lists move a
    # fifth of what the union of bricks does, and both a fraction of the dense grid
    m = sh.exchange_model(262144, 1, 8, 40520, 6390, 256 ** 3)
    assert m["brick_lists"]["bytes_per_link"] * 5 < m["union_reduce"]["bytes_per_link"] < m["dense_reduce"]["bytes_per_link"] / 4
    assert m["brick_lists"]["model_us"] < m["union_reduce"]["model_us"] < m["dense_reduce"]["model_us"]
    one = sh.exchange_model(32768, 1, 1, 6000, 6000, 128 ** 3)
    assert one["brick_lists"]["bytes_per_link"] == 0 and one["dense_reduce"]["bytes_per_link"] == 0
    # the constants are part of the result: assumed unless measured ones are handed in (bench.py measures them over the communicator at set-up)
    assert m["constants"]["source"] == "assumed" and m["constants"]["link_gbs"] == sh.XGMI_LINK_GBS and m["constants"]["latency_us"] == sh.COLLECTIVE_LATENCY_US
    mm = sh.exchange_model(262144, 1, 8, 40520, 6390, 256 ** 3, link_gbs=50.0, latency_us=10.0)
    assert mm["constants"] == {"link_gbs": 50.0, "latency_us": 10.0, "assumed": {"link_gbs": 100.0, "latency_us": 30.0}, "source": "measured at set-up (measure_p2p)"}
    assert mm["brick_lists"]["bytes_per_link"] == m["brick_lists"]["bytes_per_link"]
    assert abs(mm["brick_lists"]["model_us"] - (10.0 + m["brick_lists"]["bytes_per_link"] / 50e3)) < 0.06
    assert abs(mm["union_reduce"]["model_us"] - (20.0 + m["union_reduce"]["bytes_per_link"] / 50e3)) < 0.06


def test_no_cpu_fallback(cpm):
    """Without a GPU the product path must fail loudly (never compute on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    B = cpm.binding
    lib = B.load_library()
    h = C.c_void_p()
    rc = lib.cpm_create(0, C.byref(h))
    assert rc == -5 and not h.value  # CPM_ERR_NO_DEVICE
    assert b"no HIP device" in lib.cpm_last_error_string(None)
    with pytest.raises(B.CpmError):
        B.Context(0)


def test_oracle_is_not_imported_by_the_product():
    pkg = next(p for p in REPO.iterdir() if p.is_dir() and p.name.endswith("_amd"))
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.cpp")):
        text = f.read_text()
        assert "oracle_binding" not in text and "cpm_oracle" not in text and "libcpm_oracle" not in text, f


def test_headers_are_plain_c_and_cxx(tmp_path):
    """include/cpm/*.h is the boundary: it must compile on its own as C99 and as C++11 (no torch, no HIP types)."""
    import shutil
    import subprocess
    from pathlib import Path
    repo = Path(__file__).resolve().parent.parent
    src = tmp_path / "hdr.c"
    src.write_text('#include <cpm/cpm.h>\n#include <cpm/cpm_ext.h>\n#include <cpm/cpm_profile.h>\n'
                   'int main(void) { cpm_trace_params p; p.max_interactions = 1; return p.max_interactions - 1 + (CPM_OK != 0); }\n')
    for cc, std, extra in (("gcc", "-std=c99", []), ("g++", "-std=c++11", ["-x", "c++"])):
        if not shutil.which(cc):
            continue
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", str(repo / "include"), "-fsyntax-only", *extra, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_fast_formulation_sizing_is_host_arithmetic(cpm):
    """cpm_fast_table_entries / cpm_gather_fast_supported / cpm_fast_record_capacity are pure host functions (what a caller sizes
    its buffers with before any launch): the table has room for the brick shape of a narrow candidate box (8 x 8 x 16 voxels) and
    for those a wide box chooses (16 voxels along axes with 5 or more candidates, at most 16 x 16 x 8); radii beyond 8 candidates
    along an axis are refused; a photon is filed under at most 8 bricks."""
    B = cpm.binding
    lib = B.load_library()
    import ctypes as C
    lib.cpm_fast_table_entries.restype = C.c_size_t
    lib.cpm_fast_record_capacity.restype = C.c_size_t

    def bricks(dims, lg):
        n = [(d + (1 << l) - 1) >> l for d, l in zip(dims, lg)]
        while n[0] * n[1] * n[2] > 8192:
            a = n.index(max(n))
            lg = list(lg); lg[a] += 1
            n = [(d + (1 << l) - 1) >> l for d, l in zip(dims, lg)]
        return n[0] * n[1] * n[2]

    for dims in ((128, 128, 128), (256, 256, 48), (256, 256, 256), (32, 32, 32), (24, 64, 64)):
        g = B.default_grid_desc(dims, 1)
        shapes = [(3, 3, 4)] + [tuple(4 if (w >> a) & 1 else 3 for a in range(3)) for w in range(0, 7)] + [(4, 4, 3)]   # (what brick_shape can give)
        most = max(bricks(dims, s) for s in shapes)
        assert lib.cpm_fast_table_entries(C.byref(g), 1000) == 2 * most + 5, dims
        r1 = 1.0 / max(dims)                       # one voxel along the longest axis
        assert lib.cpm_gather_fast_supported(C.byref(g), C.c_float(0.5 * r1)) == 1
        assert lib.cpm_gather_fast_supported(C.byref(g), C.c_float(2.8 * r1)) == 1     # 6 candidates: the wide kernels
        assert lib.cpm_gather_fast_supported(C.byref(g), C.c_float(4.6 * r1)) == 0     # 10 candidates: cpm_bin + cpm_gather
        assert lib.cpm_gather_fast_supported(C.byref(g), C.c_float(0.0)) == 0
        assert lib.cpm_fast_record_capacity(C.byref(g), 1000, C.c_float(2.8 * r1)) == 1000     # a wide box is filed once
        assert lib.cpm_fast_record_capacity(C.byref(g), 1000, C.c_float(1.2 * r1)) == 8000     # 3 candidates: a record per brick touched
        assert lib.cpm_fast_record_capacity(C.byref(g), 1000, C.c_float(1.8 * r1)) == 1000     # 4: filed once
        assert lib.cpm_fast_record_capacity(C.byref(g), 1000, C.c_float(4.6 * r1)) == 0
    g = B.default_grid_desc((400, 8, 16), 1)       # reach 1 along every axis: one record per photon
    assert lib.cpm_fast_record_capacity(C.byref(g), 1000, C.c_float(0.4 / 400)) == 1000


def test_trace_lights_order_samples_is_host_arithmetic(cpm):
    """cpm_trace_lights_order_samples: the samples a cpm_trace_order for a launch over several lights is created for -- every light's
    count rounded up to whole 256-sample chunks (no device needed)."""
    import ctypes as C
    B = cpm.binding
    lib = B.load_library()
    lib.cpm_trace_lights_order_samples.restype = C.c_int
    lib.cpm_trace_lights_order_samples.argtypes = [C.c_void_p, C.c_int]
    for counts in ([1048576, 1048576], [1], [255, 257, 0, 256], [10000, 1369, 4096]):
        arr = (B.LightSpan * len(counts))()
        for a, n in zip(arr, counts):
            a.n_light_samples = n
        assert lib.cpm_trace_lights_order_samples(C.cast(arr, C.c_void_p), len(counts)) == 256 * sum((n + 255) // 256 for n in counts)
    assert lib.cpm_trace_lights_order_samples(None, 3) == 0
