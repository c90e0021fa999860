"""Mirror of kiss_icp.voxelization (src/kiss-icp/python/kiss_icp/voxelization.py:27-39).

``voxel_down_sample(points, voxel_size)`` keeps the FIRST point of every voxel,
voxel = trunc(xyz / voxel_size) per axis (Eigen ``cast<int>``, Preprocessing.cpp:58), computed on the
GPU (csrc/voxel.hip, row F1).  The reference emits survivors in tsl::robin_map iteration order
(Preprocessing.cpp:64-69); this build emits them in input order (documented deviation):
the SET of survivors is identical.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def first_per_voxel(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1) -> np.ndarray:
    """Indices (ascending = input order) of the first ``max_per_voxel`` points of every voxel."""
    xyz = torch.from_numpy(np.ascontiguousarray(points[:, :3], dtype=np.float64)).cuda()
    return ops.voxel_first(xyz, voxel_size, max_per_voxel).cpu().numpy()


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("Invalid shape")  # voxelization.py:37
    if len(points) == 0:
        return np.zeros((0, points.shape[1]), dtype=np.float64)
    return np.asarray(points[first_per_voxel(points, voxel_size, 1)], dtype=np.float64)
