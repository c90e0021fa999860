"""Mirror of kiss_icp.registration.register_frame (src/kiss-icp/python/kiss_icp/registration.py:28-73)
for 3-D point clouds: robust point-to-point Gauss-Newton ICP against a VoxelHashMap
(kiss_icp::RegisterFrame, Registration.cpp:145-195; called at registration_node.py:338-344 with
max_correspondance_distance = 3 sigma, kernel = sigma / 3).

Per iteration the GPU moves the source points by the previous update and finds the nearest map point of each in the 27
surrounding voxels (csrc/icp.hip: icp_nearest_kernel, one launch) and reduces the 6x6 normal equations
(icp_system_kernel); the host solves the 6x6 system and applies Sophus' SE3 exponential, as Registration.cpp:176-181.
Termination: |dx| < 1e-4 or 1000 iterations (Registration.cpp:92-93, 183).
The VFM-seeded 387-column variant (Registration.cpp:197-382) is not used by the headline evaluation
(every call site passes [:, :3], registration_node.py:646, 929) and is not built.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops

MAX_NUM_ITERATIONS = 1000      # Registration.cpp:92
ESTIMATION_THRESHOLD = 0.0001  # Registration.cpp:93


def se3_exp(dx: np.ndarray) -> np.ndarray:
    """Sophus::SE3d::exp, tangent = [upsilon (translation), omega (rotation)] -> 4x4 matrix."""
    ups, om = np.asarray(dx[:3], np.float64), np.asarray(dx[3:], np.float64)
    th = float(np.linalg.norm(om))
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + Om
        V = np.eye(3) + 0.5 * Om
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * (Om @ Om)
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * (Om @ Om)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


class VoxelGridDevice:
    """Sorted-key CSR copy of a VoxelHashMap's 3-D points on the GPU (what GetCorrespondences reads)."""

    def __init__(self, points: np.ndarray, voxel_size: float):
        pts = np.ascontiguousarray(points[:, :3], dtype=np.float64)
        v = np.trunc(pts / voxel_size).astype(np.int64)
        # the CSR key packs 21 bits per axis (csrc/icp.hip voxel_key); the 27-neighbour scan reaches v +- 1
        if len(v) and (np.abs(v).max() >= (1 << 20) - 1):
            raise ValueError("voxel coordinate outside +-2^20 voxels: shift the clouds towards the origin "
                             "(the ICP grid key holds 21 bits per axis)")
        v = v + (1 << 20)
        keys = (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
        order = np.argsort(keys, kind="stable")   # points of a voxel keep their insertion order
        ks = keys[order]
        uniq, first = np.unique(ks, return_index=True)
        self.voxel_size = float(voxel_size)
        self.n_voxels = len(uniq)
        self.keys = torch.from_numpy(np.ascontiguousarray(uniq)).cuda()
        self.start = torch.from_numpy(np.r_[first, len(ks)].astype(np.int32)).cuda()
        self.pts = torch.from_numpy(np.ascontiguousarray(pts[order])).cuda()


def register_frame(points: np.ndarray, voxel_map, initial_guess: np.ndarray, max_correspondance_distance: float,
                   kernel: float, src_=None, tgt_=None) -> np.ndarray:
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("Invalid shape")  # registration.py:43
    if points.shape[1] != 3:
        raise NotImplementedError("descriptor-seeded ICP (Registration.cpp:197-423) is outside the evaluated path")
    initial_guess = np.ascontiguousarray(initial_guess, dtype=np.float64)
    if voxel_map.empty():
        return initial_guess  # Registration.cpp:150
    lib = _lib.load()
    st = ops._stream()
    grid = getattr(voxel_map, "_icp_grid", None)
    if grid is None or grid[0] != len(voxel_map.point_cloud()):
        grid = (len(voxel_map.point_cloud()), VoxelGridDevice(voxel_map.point_cloud(), voxel_map.voxel_size))
        voxel_map._icp_grid = grid
    g = grid[1]
    src = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).cuda()
    n = src.shape[0]
    source = torch.empty_like(src)
    tgt = torch.empty_like(src)
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(43, dtype=torch.float64, device="cuda")
    out_h = torch.empty(43, dtype=torch.float64).pin_memory()
    # An iteration = ONE launch that moves the points by the previous update (Equation (12); the first one by the initial guess,
    # Equation (9)) and finds their nearest map points (Equation (10)), ONE that reduces the normal equations (Equation (11)),
    # and one 344-byte read-back into pinned memory; the 6x6 solve and SE3::exp stay host code as in Registration.cpp:176-177.
    # (Round 2 ran three launches, a blocking .cpu() and an upload of the 4x4 per iteration.)
    step = initial_guess
    cur = src
    T_icp = np.eye(4)
    for _ in range(MAX_NUM_ITERATIONS):
        Th = np.ascontiguousarray(step, dtype=np.float64)
        _lib.check(lib.vfm_icp_step_nearest(cur.data_ptr(), n, Th.ctypes.data, source.data_ptr(), g.keys.data_ptr(),
                                            g.start.data_ptr(), g.pts.data_ptr(), g.n_voxels, g.voxel_size,
                                            float(max_correspondance_distance), tgt.data_ptr(), valid.data_ptr(), st), "icp_step_nearest")
        cur = source
        _lib.check(lib.vfm_icp_build_system(source.data_ptr(), tgt.data_ptr(), valid.data_ptr(), n, float(kernel),
                                            out.data_ptr(), st), "icp_build_system")  # Equation (11)
        out_h.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        o = out_h.numpy()
        if o[42] == 0:
            print("[3D] No correspondences found")  # Registration.cpp:166
            break
        dx = np.linalg.solve(o[:36].reshape(6, 6), -o[36:42])  # JTJ.ldlt().solve(-JTr)
        estimation = se3_exp(dx)
        step = estimation                                       # applied by the next iteration's launch
        T_icp = estimation @ T_icp
        if np.linalg.norm(dx) < ESTIMATION_THRESHOLD:
            break
    return T_icp @ initial_guess
