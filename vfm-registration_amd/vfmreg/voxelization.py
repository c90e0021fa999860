"""Mirror of kiss_icp.voxelization (src/kiss-icp/python/kiss_icp/voxelization.py:27-39).

``voxel_down_sample(points, voxel_size)`` keeps the FIRST point of every voxel,
voxel = trunc(xyz / voxel_size) per axis (Eigen ``cast<int>``, Preprocessing.cpp:58).  The
reference emits survivors in tsl::robin_map iteration order (Preprocessing.cpp:64-69); this build
emits them in input order (documented deviation): the SET of survivors is identical.

Row F1 of SURVEY.md section 8 ("next"): this step runs BEFORE the hot path and is host-side code in
the reference too (C++ hash map on the CPU); it is vectorised numpy here and slated for a HIP
sort-by-key kernel.  It is not part of the measured path.
"""
from __future__ import annotations

import numpy as np


def voxel_keys(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """int64 key per point from the truncated voxel coordinates (21 bits per axis)."""
    v = np.trunc(np.asarray(points[:, :3], dtype=np.float64) / voxel_size).astype(np.int64)
    if np.abs(v).max(initial=0) >= (1 << 20):
        raise ValueError("voxel coordinate out of range")
    v += 1 << 20
    return (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]


def first_per_voxel(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1) -> np.ndarray:
    """Indices (ascending = input order) of the first ``max_per_voxel`` points of every voxel."""
    n = len(points)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    keys = voxel_keys(points, voxel_size)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    start = np.r_[True, ks[1:] != ks[:-1]]
    run_start = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    rank = np.arange(n) - run_start          # occurrence number inside the voxel, in input order
    keep = order[rank < max_per_voxel]
    keep.sort()
    return keep


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("Invalid shape")  # voxelization.py:37
    return np.asarray(points[first_per_voxel(points, voxel_size, 1)], dtype=np.float64)
