"""UniformGrid3D sequence files (.u3d text header + .raw payload), numpy side.

Same format as the C++ reader/writer of the host layer (host/cpm_timevarying.cpp), which follow
ref uniformgridcl/uniformgrid3dwriter.cpp:47-102 and uniformgridcl/uniformgrid3dreader.cpp:59-183:

    RawFile: <stem>.raw
    Resolution: x y z t
    Format: Vec2UINT16 | FLOAT32
    ModelMatrix: 16 floats, row by row
    WorldMatrix: 16 floats, row by row
    CellDimensions: cx cy cz

The raw file holds the t elements back to back, x fastest.  The reader also accepts the
keys ObjectFileName / Dimensions, '#' and '/' comment lines, and any key case.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

FORMATS = {"Vec2UINT16": (np.uint16, 2), "FLOAT32": (np.float32, 1)}


@dataclass
class GridSequence:
    data: np.ndarray                       # [t, z, y, x] (FLOAT32) or [t, z, y, x, 2] (Vec2UINT16)
    cell_dimensions: tuple = (8, 8, 8)
    model_matrix: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))  # row-major, as math
    world_matrix: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))

    @property
    def format(self) -> str:
        return "Vec2UINT16" if self.data.dtype == np.uint16 else "FLOAT32"


def write(path: str, seq: GridSequence, overwrite: bool = True) -> None:
    if seq.data.shape[0] < 1:
        raise ValueError("Error: Cannot write empty vector")
    raw = os.path.splitext(path)[0] + ".raw"
    if not overwrite and (os.path.exists(path) or os.path.exists(raw)):
        raise FileExistsError(path)
    t, z, y, x = seq.data.shape[:4]
    mat = lambda m: " ".join(repr(float(np.float32(v))) for v in np.asarray(m, np.float32).reshape(16))
    with open(path, "w") as f:
        f.write(f"RawFile: {os.path.splitext(os.path.basename(path))[0]}.raw\n")
        f.write(f"Resolution: {x} {y} {z} {t}\n")
        f.write(f"Format: {seq.format}\n")
        f.write(f"ModelMatrix: {mat(seq.model_matrix)}\n")
        f.write(f"WorldMatrix: {mat(seq.world_matrix)}\n")
        f.write("CellDimensions: {} {} {}\n".format(*seq.cell_dimensions))
    np.ascontiguousarray(seq.data).tofile(raw)


def read(path: str) -> GridSequence:
    raw = None
    res = None
    fmt = None
    model = np.eye(4, dtype=np.float32)
    world = np.eye(4, dtype=np.float32)
    cell = (0, 0, 0)
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line[0] in "#/":
                continue
            parts = line.split("#")[0].split(":")
            if len(parts) != 2:
                continue
            key, value = parts[0].strip().lower(), parts[1].strip()
            if key in ("objectfilename", "rawfile"):
                raw = os.path.join(os.path.dirname(path), value)
            elif key in ("resolution", "dimensions"):
                res = [int(v) for v in value.split()]
            elif key == "format":
                fmt = value.split()[0]
            elif key == "modelmatrix":
                model = np.array(value.split(), np.float32).reshape(4, 4)
            elif key == "worldmatrix":
                world = np.array(value.split(), np.float32).reshape(4, 4)
            elif key == "celldimensions":
                cell = tuple(int(v) for v in value.split())
    if not res:
        raise ValueError(f'Error: Unable to find "Resolution" tag in file: {path}')
    if fmt is None:
        raise ValueError(f'Error: Unable to find "Format" tag in file: {path}')
    if fmt not in FORMATS:
        raise ValueError(f"Error: Unsupported data format {fmt} in {path}")
    dt, comps = FORMATS[fmt]
    x, y, z, t = res
    n = x * y * z * t * comps
    data = np.fromfile(raw, dtype=dt, count=n)
    if data.size != n:
        raise ValueError(f"Error: raw file is too short: {raw}")
    shape = (t, z, y, x) + ((2,) if comps == 2 else ())
    return GridSequence(data.reshape(shape), cell, model, world)
