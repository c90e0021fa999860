"""Mirror of kiss_icp.registration.register_frame (src/kiss-icp/python/kiss_icp/registration.py:28-73)
for 3-D point clouds: robust point-to-point Gauss-Newton ICP against a VoxelHashMap
(kiss_icp::RegisterFrame, Registration.cpp:145-195; called at registration_node.py:338-344 with
max_correspondance_distance = 3 sigma, kernel = sigma / 3).

Per iteration the GPU moves the source points by the previous update and finds the nearest map point of each in the 27
surrounding voxels (csrc/icp.hip: icp_nearest_kernel, one launch) and reduces the 6x6 normal equations
(icp_system_kernel); the host solves the 6x6 system and applies Sophus' SE3 exponential, as Registration.cpp:176-181.
Termination: |dx| < 1e-4 or 1000 iterations (Registration.cpp:92-93, 183).
The descriptor-seeded variant for rows that carry descriptors (Registration.cpp:197-382; not on the headline evaluation: every call
site there passes [:, :3], registration_node.py:646, 929) is ``register_frame`` on 3 + D columns (round 4): 5 m subset ->
GetVFMCorrespondences(0.8) on the GPU -> Gauss-Newton on those pairs with median + 1.5 MAD pruning -> the vanilla loop.
Rows of any OTHER width > 3 take RegisterFrame(std::vector<Eigen::VectorXd> ...) (Registration.cpp:384-423; round 6): the 3-D loop whose
search weighs the squared distance of a neighbour by the cosine distance of the descriptors (VoxelHashMap.cpp:321-448).
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


def _grid_of(voxel_map) -> "VoxelGridDevice":
    """the sorted-key CSR of the map's 3-D points, kept with the map (VoxelHashMap.add_points drops it)"""
    grid = getattr(voxel_map, "_icp_grid", None)
    if grid is None or grid[0] != len(voxel_map.point_cloud()):
        grid = (len(voxel_map.point_cloud()), VoxelGridDevice(voxel_map.point_cloud(), voxel_map.voxel_size))
        voxel_map._icp_grid = grid
    return grid[1]


def _icp_loop(cur: torch.Tensor, g: "VoxelGridDevice", first_step, max_correspondance_distance: float, kernel: float,
              max_iters: int, moved_with=None, T_start=None):
    """The point-to-point loop of Registration.cpp:159-190 / :347-372 on device-resident points.  ``first_step``: a 4x4 applied to
    ``cur`` by the first launch (the 3-D path's initial guess, Equation (9)), or None (points already in place).
    An iteration = ONE launch that moves the points by the previous update (Equation (12)) and finds their nearest map points
    (Equation (10)), ONE that reduces the normal equations (Equation (11)), and one 344-byte read-back into pinned memory; the
    6x6 solve and SE3::exp stay host code as in Registration.cpp:176-177.  ``moved_with``: a device [k, 3] array every update is
    applied to as well (Registration.cpp:366); ``T_start``: the product of the updates so far (every update multiplies it from the
    left, in order, as Registration.cpp:181 / :363 do).  Returns (T_icp, moved_with, iterations that ran to their end).
    ``cur`` is updated in place."""
    lib = _lib.load()
    st = ops._stream()
    n = cur.shape[0]
    source = torch.empty_like(cur)
    tgt = torch.empty_like(cur)
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(43, dtype=torch.float64, device="cuda")
    out_h = torch.empty(43, dtype=torch.float64).pin_memory()
    step = first_step
    T_icp = np.eye(4) if T_start is None else np.array(T_start, dtype=np.float64)
    done = 0
    for _ in range(max_iters):
        if step is None:
            _lib.check(lib.vfm_icp_nearest(cur.data_ptr(), n, g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(), g.n_voxels,
                                           g.voxel_size, float(max_correspondance_distance), tgt.data_ptr(), valid.data_ptr(), st), "icp_nearest")
            source = cur
        else:
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
        dx = _solve6(o[:36].reshape(6, 6), -o[36:42])  # JTJ.ldlt().solve(-JTr)
        if dx is None:
            print("[3D] singular system")                       # (fewer than three non-collinear pairs: no update can be estimated)
            break
        estimation = se3_exp(dx)
        step = estimation                                       # applied by the next iteration's launch (in place from then on)
        T_icp = estimation @ T_icp
        if moved_with is not None and len(moved_with):
            moved_with = ops.transform_xyz(moved_with, torch.from_numpy(np.ascontiguousarray(estimation)).cuda())
        if np.linalg.norm(dx) < ESTIMATION_THRESHOLD:
            break
        done += 1
    return T_icp, moved_with, done


def _median_like_the_reference(v: np.ndarray) -> float:
    """Registration.cpp:297-309: nth_element at size / 2; for an even size the mean of it and the largest element before it"""
    s = np.sort(v)
    n = len(s) // 2
    return float(s[n]) if len(s) & 1 else float((s[n] + s[n - 1]) / 2)


EUCL_DIST_THRESHOLD = 0.01  # Registration.cpp:94


def _solve6(JTJ: np.ndarray, rhs: np.ndarray):
    """JTJ.ldlt().solve(rhs) (Registration.cpp:261) where the system is regular -- numpy's LU, the oracle's arithmetic --, None where it is
    singular or the solution is not finite."""
    try:
        dx = np.linalg.solve(JTJ, rhs)
    except np.linalg.LinAlgError:
        return None
    return dx if np.isfinite(dx).all() else None


def _register_frame_nd(points: np.ndarray, voxel_map, initial_guess: np.ndarray, max_correspondance_distance: float, kernel: float):
    """Registration.cpp:197-382 (see the module docstring); returns (pose, src_, tgt_)."""
    from .voxelization import down_sample_device, to_device_rows
    lib = _lib.load()
    st = ops._stream()
    rows, xyz = to_device_rows(points)
    T0 = torch.from_numpy(initial_guess).cuda()
    source_3d = ops.transform_xyz(xyz, T0)                                      # :207-208 (descriptors carried through)
    _, sub_xyz, order = down_sample_device(rows, source_3d, 5.0)               # :216
    if len(order) < 100:
        print("[WARNING] Voxelized too sparse. Keep input.")                   # :217-220
        order = torch.arange(len(rows), device=rows.device)
        sub_xyz = source_3d
    qi, mi, _ = voxel_map.search_device(rows[order], 0.8)                      # :229-230 (only the descriptor columns matter)
    src_3d = sub_xyz[qi].contiguous()
    tgt_3d = voxel_map._device_map()[1][mi].contiguous()

    def dists(a: np.ndarray, b: np.ndarray) -> np.ndarray:                     # (host fp64, the oracle's expression order)
        d = a - b
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    tgt_h = tgt_3d.cpu().numpy()
    prev = float(np.sum(dists(src_3d.cpu().numpy(), tgt_h)) / len(tgt_h)) if len(tgt_h) else float("nan")   # :233-240
    out = torch.empty(43, dtype=torch.float64, device="cuda")
    out_h = torch.empty(43, dtype=torch.float64).pin_memory()
    T_icp = np.eye(4)
    j = 0
    while j < MAX_NUM_ITERATIONS:                                               # :253
        k = src_3d.shape[0]
        if k == 0:
            print("No correspondences found")
            break
        ones = torch.ones(k, dtype=torch.uint8, device="cuda")
        _lib.check(lib.vfm_icp_build_system(src_3d.data_ptr(), tgt_3d.data_ptr(), ones.data_ptr(), k, float(kernel), out.data_ptr(), st),
                   "icp_build_system")
        out_h.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        o = out_h.numpy()
        dx = _solve6(o[:36].reshape(6, 6), -o[36:42])
        if dx is None:
            # one or two surviving pairs, or collinear ones: the 6 x 6 system is singular.  The reference's JTJ.ldlt().solve() does not
            # throw there (its update is then arbitrary); here the descriptor-seeded loop stops and the vanilla loop below takes over
            # from the pose reached so far (numpy's LinAlgError used to escape)
            print("[ND] singular system in the VFM loop")
            break
        est = se3_exp(dx)
        est_d = torch.from_numpy(np.ascontiguousarray(est)).cuda()
        source_3d = ops.transform_xyz(source_3d, est_d)                        # :265-266
        src_3d = ops.transform_xyz(src_3d, est_d)
        T_icp = est @ T_icp
        d = dists(src_3d.cpu().numpy(), tgt_h)
        mean = float(np.add.accumulate(d)[-1] / len(d))                         # std::accumulate: in order
        median = _median_like_the_reference(d)
        mad = _median_like_the_reference(np.abs(d - median)) * 1.4826          # :311-322
        keep = np.abs(d - median) < 1.5 * mad                                  # :326-331
        if not keep.all():
            kd = torch.from_numpy(np.flatnonzero(keep)).cuda()
            src_3d, tgt_3d = src_3d[kd].contiguous(), tgt_3d[kd].contiguous()
            tgt_h = tgt_h[keep]
        if abs(prev - mean) < EUCL_DIST_THRESHOLD:                             # :332-334 (j is not incremented on break)
            break
        prev = mean
        j += 1
    print(f"[ND] [{j}] finished VFM")
    # vanilla loop on all points, iteration counter carried over (:347-372); src_ moves with every update (:366)
    g = _grid_of(voxel_map if not voxel_map.empty() else voxel_map.xyz_map())
    T_icp, src_moved, done = _icp_loop(source_3d, g, None, max_correspondance_distance, kernel, MAX_NUM_ITERATIONS - j, moved_with=src_3d,
                                       T_start=T_icp)
    print(f"[ND] [{j + done}] finished")
    pose = T_icp @ initial_guess
    return pose, src_moved.cpu().numpy(), tgt_3d.cpu().numpy()


POINT_SIZE = 3 + 384   # kiss_icp_pybind._point_size(): 3 + DESCRIPTOR_SIZE, a compile-time 384 (DescriptorSize.hpp:7) -- the width that
                       # takes the descriptor-seeded RegisterFrame; every other width > 3 takes the VectorXd one (registration.py:37-42)


class _DescGrid:
    """The ICP grid of a map's rows WITH their descriptor columns (what VoxelHashMap::GetCorrespondences(VectorXdVector) reads from
    map_x_): the CSR of ``VoxelGridDevice`` + the descriptor rows in its order as fp64 + their norms / "element sum != 0" flags."""

    def __init__(self, rows: np.ndarray, voxel_size: float):
        lib = _lib.load()
        rows = np.asarray(rows, dtype=np.float64)
        pts = np.ascontiguousarray(rows[:, :3])
        v = np.trunc(pts / voxel_size).astype(np.int64)
        if len(v) and (np.abs(v).max() >= (1 << 20) - 1):
            raise ValueError("voxel coordinate outside +-2^20 voxels: shift the clouds towards the origin")
        v = v + (1 << 20)
        keys = (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
        order = np.argsort(keys, kind="stable")
        uniq, first = np.unique(keys[order], return_index=True)
        self.voxel_size = float(voxel_size)
        self.n_voxels = len(uniq)
        self.f = rows.shape[1] - 3
        self.keys = torch.from_numpy(np.ascontiguousarray(uniq)).cuda()
        self.start = torch.from_numpy(np.r_[first, len(order)].astype(np.int32)).cuda()
        self.pts = torch.from_numpy(np.ascontiguousarray(pts[order])).cuda()
        self.desc = torch.from_numpy(np.ascontiguousarray(rows[order, 3:])).cuda()
        self.norm = torch.empty(len(order), dtype=torch.float64, device="cuda")
        self.has = torch.empty(len(order), dtype=torch.uint8, device="cuda")
        _lib.check(lib.vfm_icp_desc_stats(self.desc.data_ptr(), len(order), self.f, self.norm.data_ptr(), self.has.data_ptr(), ops._stream()),
                   "icp_desc_stats")


def _register_frame_xd(points: np.ndarray, voxel_map, initial_guess: np.ndarray, max_correspondance_distance: float, kernel: float):
    """RegisterFrame(std::vector<Eigen::VectorXd> ...) (Registration.cpp:384-423): the 3-D loop with the descriptor-weighted nearest
    neighbour of VoxelHashMap.cpp:321-448 as its search (csrc/icp.hip icp_nearest_desc_kernel).  Every iterate equals the oracle's
    (oracle.register_frame_xd) bit for bit."""
    lib = _lib.load()
    st = ops._stream()
    g = getattr(voxel_map, "_icp_desc_grid", None)
    rows_n = voxel_map._cloud("n")
    if g is None or g[0] != len(rows_n[0]):
        g = (len(rows_n[0]), _DescGrid(voxel_map.point_cloud_n(), voxel_map.voxel_size))
        voxel_map._icp_desc_grid = g
    g = g[1]
    pts = np.asarray(points, dtype=np.float64)
    n, f = pts.shape[0], pts.shape[1] - 3
    cur = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).cuda()
    sdesc = torch.from_numpy(np.ascontiguousarray(pts[:, 3:])).cuda()
    snorm = torch.empty(n, dtype=torch.float64, device="cuda")
    shas = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_icp_desc_stats(sdesc.data_ptr(), n, f, snorm.data_ptr(), shas.data_ptr(), st), "icp_desc_stats")
    source = torch.empty_like(cur)
    tgt = torch.empty_like(cur)
    valid = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(43, dtype=torch.float64, device="cuda")
    out_h = torch.empty(43, dtype=torch.float64).pin_memory()
    step = np.ascontiguousarray(initial_guess, dtype=np.float64)    # Equation (9), applied by the first launch
    T_icp = np.eye(4)
    for j in range(MAX_NUM_ITERATIONS):
        Th = np.ascontiguousarray(step, dtype=np.float64)
        _lib.check(lib.vfm_icp_step_nearest_desc(cur.data_ptr(), n, Th.ctypes.data, source.data_ptr(), sdesc.data_ptr(), snorm.data_ptr(),
                                                 shas.data_ptr(), f, g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(),
                                                 g.desc.data_ptr(), g.norm.data_ptr(), g.has.data_ptr(), g.n_voxels, g.voxel_size,
                                                 float(max_correspondance_distance), tgt.data_ptr(), valid.data_ptr(), st),
                   "icp_step_nearest_desc")
        cur = source
        _lib.check(lib.vfm_icp_build_system(source.data_ptr(), tgt.data_ptr(), valid.data_ptr(), n, float(kernel), out.data_ptr(), st),
                   "icp_build_system")
        out_h.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        o = out_h.numpy()
        if j == 0:
            print(f"[XD] [{j}] correspondences {int(o[42])}")          # Registration.cpp:400
        if o[42] == 0:
            break
        dx = _solve6(o[:36].reshape(6, 6), -o[36:42])
        if dx is None:
            break
        estimation = se3_exp(dx)
        step = estimation
        T_icp = estimation @ T_icp
        if np.linalg.norm(dx) < ESTIMATION_THRESHOLD:
            break
    return T_icp @ initial_guess


def register_frame(points: np.ndarray, voxel_map, initial_guess: np.ndarray, max_correspondance_distance: float,
                   kernel: float, src_=None, tgt_=None):
    points = np.asarray(points)
    if points.ndim != 2 or points.shape[1] < 3:
        raise ValueError("Invalid shape")  # registration.py:43
    initial_guess = np.ascontiguousarray(initial_guess, dtype=np.float64)
    if points.shape[1] != 3:
        # registration.py:37-66: rows of _point_size() = 387 columns take the descriptor-seeded RegisterFrame and return the surviving pairs
        # when the caller passed src_ / tgt_; every other width > 3 takes RegisterFrame(VectorXd ...) (Registration.cpp:384-423, round 6):
        # the 3-D loop with the descriptor-weighted nearest neighbour as its search, against the map's rows of the same width (the
        # reference keeps those in map_x_; here the wide rows of a map live in one container whatever their width)
        ncols = None if voxel_map.empty_n() else voxel_map._cloud("n")[0].shape[1]
        if ncols is not None and points.shape[1] != ncols:
            raise ValueError("Invalid shape")   # (rows of a width the map does not hold: the reference's map_x_ would be empty or of another width)
        if points.shape[1] != POINT_SIZE:
            if voxel_map.empty_n():
                return initial_guess                                # Registration.cpp:389
            return _register_frame_xd(points, voxel_map, initial_guess, max_correspondance_distance, kernel)
        if voxel_map.empty_n():
            pose, s_out, t_out = initial_guess, np.asarray(src_ if src_ is not None else [[0, 0, 0]], dtype=np.float64), \
                np.asarray(tgt_ if tgt_ is not None else [[0, 0, 0]], dtype=np.float64)   # Registration.cpp:204
        else:
            pose, s_out, t_out = _register_frame_nd(points, voxel_map, initial_guess, max_correspondance_distance, kernel)
        return pose if (src_ is None or tgt_ is None) else (pose, s_out, t_out)
    if voxel_map.empty():
        return initial_guess  # Registration.cpp:150
    src = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float64)).cuda()
    T_icp, _, _ = _icp_loop(src, _grid_of(voxel_map), initial_guess, max_correspondance_distance, kernel, MAX_NUM_ITERATIONS)
    return T_icp @ initial_guess
