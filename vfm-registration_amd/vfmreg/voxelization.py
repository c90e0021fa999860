"""Mirror of kiss_icp.voxelization (src/kiss-icp/python/kiss_icp/voxelization.py:27-39).

``voxel_down_sample(points, voxel_size)`` keeps the FIRST point of every voxel,
voxel = trunc(xyz / voxel_size) per axis (Eigen ``cast<int>``, Preprocessing.cpp:58), and returns the
survivors in the order the reference does: the iteration order of the ``tsl::robin_map`` that
``VoxelDownsample`` fills (``reserve(frame.size())``, Preprocessing.cpp:55-69).  The order matters
because the path chains the call (registration_node.py:399-414: 0.5 m -> 1.0 m -> 5.0 m): each level
keeps the first point per voxel OF THE PREVIOUS LEVEL'S OUTPUT ORDER.  Both parts run on the GPU
(csrc/voxel.hip, row F1).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def to_device_rows(points: np.ndarray):
    """One upload of a point block in its own dtype + the fp64 coordinates the containers key on (the pybind layer
    force-casts every input to double, stl_vector_eigen.h:73-86; fp32 -> fp64 is exact, so the descriptors can stay in
    the given dtype until they are cast to float for the search, VoxelHashMap.cpp:472-481)."""
    rows = torch.from_numpy(np.ascontiguousarray(points)).cuda()
    if rows.dtype not in (torch.float32, torch.float64):
        rows = rows.double()
    return rows, rows[:, :3].double().contiguous()


def down_sample_device(rows: torch.Tensor, xyz64: torch.Tensor, voxel_size: float):
    """voxel_down_sample on device-resident rows: (rows[order], xyz64[order], order) in the container's order."""
    order = ops.voxel_robin(xyz64, voxel_size)
    return rows[order], xyz64[order], order


def first_per_voxel(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1) -> np.ndarray:
    """Indices (ascending = input order) of the first ``max_per_voxel`` points of every voxel."""
    xyz = torch.from_numpy(np.ascontiguousarray(points[:, :3], dtype=np.float64)).cuda()
    return ops.voxel_first(xyz, voxel_size, max_per_voxel).cpu().numpy()


def robin_order(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1, reserve: bool = True,
                hash_mul: int = ops.HASH_DOWNSAMPLE) -> np.ndarray:
    """Indices of the kept points in the reference's container iteration order."""
    xyz = torch.from_numpy(np.ascontiguousarray(points[:, :3], dtype=np.float64)).cuda()
    return ops.voxel_robin(xyz, voxel_size, max_per_voxel, reserve, hash_mul).cpu().numpy()


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("Invalid shape")  # voxelization.py:37
    if len(points) == 0:
        return np.zeros((0, points.shape[1]), dtype=np.float64)
    rows, xyz64 = to_device_rows(points)
    kept, _, _ = down_sample_device(rows, xyz64, voxel_size)   # gather on the device: only the survivors come back
    return kept.cpu().numpy().astype(np.float64, copy=False)
