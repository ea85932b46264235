"""The .u3d grid-sequence format: C++ host reader/writer (libcpm_host) and the numpy side agree and
round-trip; header dialects and error cases of the reference reader.  No GPU needed."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def host(cpm):
    lib = C.CDLL(str(cpm.binding.LIB_PATH.parent / "libcpm_host.so"))
    lib.cpmh_last_error.restype = C.c_char_p
    lib.cpmh_u3d_write.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int * 3), C.POINTER(C.c_int * 3), C.POINTER(C.c_float * 16),
                                   C.POINTER(C.c_float * 16), C.c_void_p, C.c_int, C.c_int]
    lib.cpmh_u3d_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int * 3), C.POINTER(C.c_int * 3),
                                  C.POINTER(C.c_float * 16), C.POINTER(C.c_float * 16), C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
    lib.cpmh_u3d_read_data.argtypes = [C.c_void_p]
    return lib


@pytest.fixture(scope="module")
def u3d(cpm):
    return importlib.import_module(cpm.__name__ + ".u3d")


def _cpp_write(host, path, data, cell, model, world, overwrite=True):
    fmt = 0 if data.dtype == np.uint16 else 1
    t, z, y, x = data.shape[:4]
    data = np.ascontiguousarray(data)
    # column-major float[16] on the C side
    m = (C.c_float * 16)(*np.asarray(model, np.float32).T.reshape(16))
    w = (C.c_float * 16)(*np.asarray(world, np.float32).T.reshape(16))
    return host.cpmh_u3d_write(str(path).encode(), fmt, C.byref((C.c_int * 3)(x, y, z)), C.byref((C.c_int * 3)(*cell)), C.byref(m),
                               C.byref(w), data.ctypes.data, t, int(overwrite))


def _cpp_read(host, path):
    fmt, count, eb = C.c_int(), C.c_int(), C.c_ulonglong()
    dims, cell = (C.c_int * 3)(), (C.c_int * 3)()
    m, w = (C.c_float * 16)(), (C.c_float * 16)()
    rc = host.cpmh_u3d_read(str(path).encode(), C.byref(fmt), C.byref(dims), C.byref(cell), C.byref(m), C.byref(w), C.byref(count),
                            C.byref(eb))
    if rc != 0:
        raise ValueError(host.cpmh_last_error().decode())
    x, y, z = dims
    shape = (count.value, z, y, x) + ((2,) if fmt.value == 0 else ())
    out = np.empty(shape, np.uint16 if fmt.value == 0 else np.float32)
    assert out.nbytes == eb.value * count.value
    assert host.cpmh_u3d_read_data(out.ctypes.data) == 0
    return out, tuple(cell), np.array(m, np.float32).reshape(4, 4).T, np.array(w, np.float32).reshape(4, 4).T


def _example(dtype, t=3, dims=(5, 4, 3)):
    rng = np.random.default_rng(7)
    x, y, z = dims
    if dtype == np.uint16:
        data = rng.integers(0, 65536, (t, z, y, x, 2)).astype(np.uint16)
    else:
        data = rng.standard_normal((t, z, y, x)).astype(np.float32)
    model = np.eye(4, dtype=np.float32)
    model[:3, :3] = np.diag([2.0, 1.5, 0.75])
    model[:3, 3] = [-1.0, -0.75, 0.125]   # translation in the last column
    world = np.eye(4, dtype=np.float32)
    world[0, 1] = 0.1
    return data, (8, 8, 4), model, world


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_cpp_round_trip_and_header(host, tmp_path, dtype):
    data, cell, model, world = _example(dtype)
    path = tmp_path / "grids.u3d"
    assert _cpp_write(host, path, data, cell, model, world) == 0
    text = path.read_text().splitlines()
    keys = [l.split(":")[0] for l in text]
    assert keys == ["RawFile", "Resolution", "Format", "ModelMatrix", "WorldMatrix", "CellDimensions"]  # writer order
    assert text[0] == "RawFile: grids.raw"
    assert text[1] == "Resolution: 5 4 3 3"
    assert text[2] == "Format: " + ("Vec2UINT16" if dtype == np.uint16 else "FLOAT32")
    assert [float(v) for v in text[3].split(":")[1].split()] == list(model.reshape(16))  # row by row
    assert (tmp_path / "grids.raw").stat().st_size == data.nbytes
    got, gcell, gm, gw = _cpp_read(host, path)
    assert np.array_equal(got, data) and gcell == cell
    assert np.array_equal(gm, model) and np.array_equal(gw, world)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_cpp_and_numpy_sides_interoperate(host, u3d, tmp_path, dtype):
    data, cell, model, world = _example(dtype, t=2)
    a, b = tmp_path / "a.u3d", tmp_path / "b.u3d"
    assert _cpp_write(host, a, data, cell, model, world) == 0
    seq = u3d.read(str(a))
    assert np.array_equal(seq.data, data) and tuple(seq.cell_dimensions) == cell
    assert np.array_equal(seq.model_matrix, model) and np.array_equal(seq.world_matrix, world)
    u3d.write(str(b), seq)
    got, gcell, gm, gw = _cpp_read(host, b)
    assert np.array_equal(got, data) and gcell == cell and np.array_equal(gm, model) and np.array_equal(gw, world)
    assert (tmp_path / "a.raw").read_bytes() == (tmp_path / "b.raw").read_bytes()


def test_reader_header_dialect(host, u3d, tmp_path):
    """Keys are case-insensitive, ObjectFileName / Dimensions are synonyms, '#' and '/' start comments,
    lines without exactly one ':' are skipped, matrices default to identity (uniformgrid3dreader.cpp:77-118)."""
    data = np.arange(2 * 2 * 3 * 4, dtype=np.float32).reshape(2, 2, 3, 4)
    data.tofile(tmp_path / "payload.bin")
    (tmp_path / "h.u3d").write_text(
        "# a comment\n// another\n\nOBJECTFILENAME: payload.bin\nDimensions: 4 3 2 2   # trailing comment\n"
        "format: FLOAT32\nthis line has no colon\nUnknownKey: 1 2 3\ncelldimensions: 16 16 16\n")
    for got, cell, m in [_cpp_read(host, tmp_path / "h.u3d")[:3],
                         (lambda s: (s.data, tuple(s.cell_dimensions), s.model_matrix))(u3d.read(str(tmp_path / "h.u3d")))]:
        assert np.array_equal(got, data) and cell == (16, 16, 16) and np.array_equal(m, np.eye(4, dtype=np.float32))


def test_reader_and_writer_errors(host, u3d, tmp_path):
    p = tmp_path / "bad.u3d"
    p.write_text("RawFile: x.raw\nFormat: FLOAT32\n")
    with pytest.raises(ValueError, match="Resolution"):
        _cpp_read(host, p)
    with pytest.raises(ValueError, match="Resolution"):
        u3d.read(str(p))
    p.write_text("RawFile: x.raw\nResolution: 2 2 2 1\n")
    with pytest.raises(ValueError, match="Format"):
        _cpp_read(host, p)
    p.write_text("RawFile: x.raw\nResolution: 2 2 2 1\nFormat: FLOAT33\n")
    with pytest.raises(ValueError, match="not a data format name"):
        _cpp_read(host, p)
    p.write_text("RawFile: x.raw\nResolution: 2 2 2 1\nFormat: UINT8\n")
    with pytest.raises(ValueError, match="not supported"):
        _cpp_read(host, p)
    p.write_text("RawFile: missing.raw\nResolution: 2 2 2 1\nFormat: FLOAT32\n")
    with pytest.raises(ValueError, match="cannot open the data file"):
        _cpp_read(host, p)
    np.zeros(7, np.float32).tofile(tmp_path / "short.raw")
    p.write_text("RawFile: short.raw\nResolution: 2 2 2 1\nFormat: FLOAT32\n")
    with pytest.raises(ValueError, match="ends before element 0 is complete"):
        _cpp_read(host, p)
    with pytest.raises(ValueError, match="too short"):
        u3d.read(str(p))
    # writer: empty vector, overwrite protection
    data, cell, model, world = _example(np.float32, t=1)
    assert _cpp_write(host, tmp_path / "w.u3d", data[:0], cell, model, world) == -1
    assert b"empty sequence" in host.cpmh_last_error()
    assert _cpp_write(host, tmp_path / "w.u3d", data, cell, model, world) == 0
    assert _cpp_write(host, tmp_path / "w.u3d", data, cell, model, world, overwrite=False) == -1
    with pytest.raises(FileExistsError):
        u3d.write(str(tmp_path / "w.u3d"), u3d.GridSequence(data), overwrite=False)
