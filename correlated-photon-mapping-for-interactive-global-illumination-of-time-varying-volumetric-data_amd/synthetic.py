"""Synthetic inputs of the BASELINE.json configurations.

Pure numpy, deterministic, no RNG: the same arrays feed the HIP path, the
oracle and the CPU baseline.  Nothing here touches a GPU.
"""
from __future__ import annotations

import math

import numpy as np

# six-point transfer function of the reference workspace
# (workspaces/CorrelatedPhotonMappingSingleVolume.inv:662-687): (position, r, g, b, a)
WORKSPACE_TF_POINTS = [
    (0.01686747, 1.0, 0.59633785, 0.24313726, 0.0),
    (0.036445361, 0.90980393, 0.49831387, 0.29256123, 0.0),
    (0.073654622, 0.93725491, 0.58783853, 0.48149845, 0.0),
    (0.22178316, 0.6156863, 0.25906768, 0.10623331, 0.18884119),
    (0.28514057, 0.93725491, 0.1506981, 0.25557336, 0.39484981),
    (0.67068273, 0.10786603, 0.61843288, 0.65490198, 0.53218883),
]


def tf_from_points(points, width: int = 1024) -> np.ndarray:
    """Piecewise-linear RGBA LUT, texel i at position (i + 0.5) / width; constant outside
    the first / last point (Inviwo TransferFunction -> 1024 x 1 RGBA32F layer)."""
    pts = sorted(points)
    pos = np.array([p[0] for p in pts], dtype=np.float64)
    col = np.array([p[1:] for p in pts], dtype=np.float64)
    x = (np.arange(width, dtype=np.float64) + 0.5) / width
    out = np.empty((width, 4), dtype=np.float64)
    for c in range(4):
        out[:, c] = np.interp(x, pos, col[:, c])
    return out.astype(np.float32)


def workspace_tf(width: int = 1024, moved_point4: float | None = None) -> np.ndarray:
    """Config 2 TF; config 3 moves TF point 4 (0-based index 3) from x = 0.2218 to 0.26."""
    pts = list(WORKSPACE_TF_POINTS)
    if moved_point4 is not None:
        p = pts[3]
        pts[3] = (moved_point4,) + p[1:]
    return tf_from_points(pts, width)


def homogeneous_tf(alpha: float = 0.25, width: int = 1024) -> np.ndarray:
    out = np.ones((width, 4), dtype=np.float32)
    out[:, 3] = np.float32(alpha)
    return out


def homogeneous_volume(dim: int = 64, value: int = 128) -> np.ndarray:
    """Config 1: dim^3 u8 volume of constant value (array index order [z, y, x])."""
    return np.full((dim, dim, dim), value, dtype=np.uint8)


def heterogeneous_volume(dim=256, blob_center=(0.5, 0.5, 0.5)) -> np.ndarray:
    """Config 2/4/5: v = clamp(0.5 + 0.25 sin(8 pi x) sin(6 pi y) sin(4 pi z)
    + 0.25 exp(-|p - c|^2 / 0.02), 0, 1) at voxel centres, quantised to u8.
    dim: an int (a cube) or (dx, dy, dz) -- the workspace's own volume is 512 x 512 x 96; the array is [z, y, x]."""
    dx, dy, dz = (dim, dim, dim) if np.isscalar(dim) else (int(dim[0]), int(dim[1]), int(dim[2]))
    cx, cy, cz = [(np.arange(d, dtype=np.float64) + 0.5) / d for d in (dx, dy, dz)]
    sx = np.sin(8 * math.pi * cx)
    sy = np.sin(6 * math.pi * cy)
    sz = np.sin(4 * math.pi * cz)
    gx = (cx - blob_center[0]) ** 2
    gy = (cy - blob_center[1]) ** 2
    gz = (cz - blob_center[2]) ** 2
    out = np.empty((dz, dy, dx), dtype=np.uint8)
    for z in range(dz):
        wave = 0.25 * sz[z] * sy[:, None] * sx[None, :]
        blob = 0.25 * np.exp(-(gz[z] + gy[:, None] + gx[None, :]) / 0.02)
        v = np.clip(0.5 + wave + blob, 0.0, 1.0)
        out[z] = np.rint(v * 255.0).astype(np.uint8)
    return out


def sequence_blob_center(step: int, n_steps: int = 32):
    """Config 5: blob centre c_t = (0.3 + 0.4 t / (n - 1), 0.5, 0.5)."""
    return (0.3 + 0.4 * step / (n_steps - 1), 0.5, 0.5)


UNIT_CUBE_VERTICES = np.array(
    [[x, y, z] for z in (0.0, 1.0) for y in (0.0, 1.0) for x in (0.0, 1.0)], dtype=np.float32)
UNIT_CUBE_AABB = (0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0)


def photon_radius_texture(volume_dims, radius_voxels: float = 1.0) -> float:
    """|indexToTexture * (r, r, r, 0)| (ref processor/progressivephotontracercl.cpp:252-254)."""
    v = np.float32(radius_voxels) / np.asarray(volume_dims, dtype=np.float32)
    return float(np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], dtype=np.float32))  # glm::length in fp32
