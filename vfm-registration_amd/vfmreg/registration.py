"""Mirror of the hot-path methods of ``RegistrationNode`` (src/vfm-reg/src/registration_node.py):

  compute_vfm_correspondences  RN:396-425      ransac_registration('vfm')  RN:273-357
  find_correspondences         RN:482-538      compute_errors              RN:997-1019
  rotation re-orthogonalisation RN:331-336

Same names, argument meaning and return values; numpy in / numpy out.  The ROS node, the baseline
descriptors, TEASER and PointDSC are out of scope.  ``run_icp=True`` runs the point-to-point ICP
refinement of registration_node.py:338-344 (row F2, vfmreg/icp.py).
"""
from __future__ import annotations

import weakref
import zlib
from typing import Optional, Tuple

import numpy as np
import torch

from . import o3d, ops
from .config import load_config
from .icp import register_frame
from .mapping import get_voxel_hash_map
from .utils import transform_pcl
from .voxelization import down_sample_device, to_device_rows, voxel_down_sample


def orthogonalize_rotation(pose: np.ndarray) -> np.ndarray:
    """RN:331-336: Newton iteration towards the closest rotation (3x3 fp64, host)."""
    pose = np.array(pose, dtype=np.float64, copy=True)
    R = pose[:3, :3]
    while np.abs(1 - np.linalg.det(R)) > 1e-12:
        R = 3 / 2 * R - 1 / 2 * R @ R.T @ R
    pose[:3, :3] = R
    return pose


def compute_errors(pose: np.ndarray, gt_pose: np.ndarray) -> Tuple[float, float]:
    """RN:997-1019: (RTE in m, RRE in degrees)."""
    rte = float(np.linalg.norm(pose[:3, 3] - gt_pose[:3, 3]))
    c = (np.trace(pose[:3, :3].T @ gt_pose[:3, :3]) - 1) / 2
    rre = float(np.abs(np.arccos(np.clip(c, -1, 1))) * 180 / np.pi)
    return rte, rre


def find_correspondences(feats0: np.ndarray, feats1: np.ndarray, n_points: int = 5000, mutual_filter: bool = True):
    """RN:482-538 (adapted from TEASER++): exact Euclidean 1-NN on the GPU instead of cKDTree.

    The rows are searched as float32 (the descriptors of this path are float32 -- image_features.py:101 --; the reference's
    cKDTree works in the caller's dtype, so float64 inputs that differ only beyond float32 precision may resolve exact ties and
    near-ties differently); the decision among float32 rows is the oracle's fp64 distance, ties to the lowest index."""
    f0 = torch.from_numpy(np.ascontiguousarray(feats0, dtype=np.float32)).cuda()
    f1 = torch.from_numpy(np.ascontiguousarray(feats1, dtype=np.float32)).cuda()
    if mutual_filter:   # RN:520-532 in one call: the reverse direction is searched only at the matched rows of feats1
        i0, i1, count = ops.match_mutual_pairs(f0, f1)
        k = int(count.item())
        return i0[:k].cpu().numpy(), i1[:k].cpu().numpy()
    # RN:505-518: the n_points pairs with the smallest distance
    nn01, d2, _ = ops.match_mutual_l2(f0, f1, mutual=False)
    nns01 = nn01.cpu().numpy()
    dists = np.sqrt(d2.cpu().numpy())
    n = min(n_points, len(dists) - 1)
    top = np.argpartition(dists, n)[:n]
    return np.arange(len(nns01))[top], nns01[top]


def _fingerprint(a: np.ndarray) -> Optional[int]:
    """A cheap host-side fingerprint of a C-contiguous array: CRC of ~4096 evenly spaced elements plus its first and last rows
    (microseconds for a 600 MB map).  None for arrays whose flat view would be a copy."""
    if not a.flags.c_contiguous or a.size == 0:
        return None
    flat = a.reshape(-1)
    step = max(1, flat.size // 4096)
    return zlib.crc32(np.ascontiguousarray(flat[::step]).tobytes() + a[0].tobytes() + a[-1].tobytes())


class MapHandle:
    """A scene's map, built once (``RegistrationNode.set_map``): the VoxelHashMap of RN:402-403 with its kept rows in the container's
    order on the device, cast and prepared for the search.  Pass it wherever a method takes ``voxel_map``: the scene loop of
    RN:556-589 builds ``local_map`` once and registers every scan of the scene against it -- with a handle nothing is rebuilt and
    nothing is guessed (no fingerprint of a 600 MB array: the caller says which map it means)."""

    __slots__ = ("voxel_hash_map", "rows", "cols", "dtype")

    def __init__(self, voxel_hash_map, shape, dtype):
        self.voxel_hash_map = voxel_hash_map
        self.rows, self.cols = int(shape[0]), int(shape[1])
        self.dtype = dtype

    def __repr__(self):
        return f"MapHandle({self.rows} x {self.cols} {self.dtype})"


class RegistrationNode:
    """The registration methods of the reference's node, without ROS (RN:44-89).

    ``cache_map`` (default OFF = the reference's behaviour, a fresh map per call): the reference registers every scan of a scene against the same ``local_map`` (built once,
    RN:556-580, used at RN:587-589), and re-builds its ``VoxelHashMap`` from that array in every call (RN:402-403).  Here the built
    map -- the kept rows in the container's order, uploaded and cast for the search -- is kept for as long as the caller passes the
    SAME array object (identity, shape, dtype) with the same fingerprint (``_fingerprint``: a CRC of ~4096 sampled elements and the
    first / last rows); a 200 000 x 387 fp64 map is 619 MB of PCIe upload and ~4 ms of container replay per call otherwise.  An
    in-place edit that misses every sampled element goes unnoticed (tests/test_gpu_api.py documents one): a caller that opts in
    promises not to edit the array in place between calls, or calls ``invalidate_map()`` after doing so.  ``evaluate_scene`` -- which
    builds the scene's map itself and never edits it -- opts in."""

    def __init__(self, config=None, ransac_iterations: int = 50000, max_correspondence_distance: float = 10000.0,
                 min_cosine_similarity: float = 0.8, cache_map: bool = False):
        self.config = config or load_config(None, None)  # RN:85
        self.ransac_iterations = ransac_iterations       # RN:326
        self.max_correspondence_distance = max_correspondence_distance  # RN:323
        self.min_cosine_similarity = min_cosine_similarity              # RN:418
        self.cache_map = bool(cache_map)
        self._map_cache = None   # (weakref to the array, (shape, dtype, fingerprint), VoxelHashMap)
        self._pose_cache = None  # (bytes of the last initial pose, its device copy): the node passes the identity in every call
        self._chain_host = None  # page-locked landing area of the voxel chain's one read-back

    def invalidate_map(self) -> None:
        """Forget the kept map (``cache_map=True``): the next call rebuilds it from the array it is handed."""
        self._map_cache = None

    def set_map(self, voxel_map) -> MapHandle:
        """Build the scene's map ONCE (RN:402-403 + the upload and the search operand) and return a handle to pass in place of the array
        -- the explicit form of what ``cache_map=True`` infers from a sampled CRC.  The array may be edited or freed
        afterwards: the handle owns device copies."""
        vm = np.asarray(voxel_map)
        if vm.ndim != 2 or vm.shape[1] < 3:
            raise ValueError("Invalid shape")
        voxel_hash_map = get_voxel_hash_map(self.config)
        voxel_hash_map.add_points(vm)
        if not voxel_hash_map.empty_n():
            voxel_hash_map._device_map()
        return MapHandle(voxel_hash_map, vm.shape, vm.dtype.str)

    def _hash_map_for(self, voxel_map):
        """The VoxelHashMap of RN:402-403 for this map array -- built, or the one built for the same array before."""
        if isinstance(voxel_map, MapHandle):
            return voxel_map.voxel_hash_map
        vm = np.asarray(voxel_map)
        key = None
        if self.cache_map and isinstance(vm, np.ndarray) and vm.ndim == 2:
            fp = _fingerprint(vm)
            key = (vm.shape, vm.dtype.str, fp) if fp is not None else None
        c = self._map_cache
        if key is not None and c is not None and c[0]() is vm and c[1] == key:
            return c[2]
        voxel_hash_map = get_voxel_hash_map(self.config)                # RN:402-403
        voxel_hash_map.add_points(vm)
        if key is not None:
            if not voxel_hash_map.empty_n():
                voxel_hash_map._device_map()                            # ordered rows + fp32 descriptors, once
            try:
                self._map_cache = (weakref.ref(vm), key, voxel_hash_map)
            except TypeError:                                            # (an array-like that cannot be weakly referenced)
                self._map_cache = None
        return voxel_hash_map

    # The steps below are registration_node.py:396-425 and 288-344 line for line, but on device-resident rows: the
    # clouds are uploaded once, every voxelisation / transform / search works on the device copies, and only what the
    # caller receives (coordinates, poses) comes back.  The numpy-level mirrors (voxel_down_sample, transform_pcl,
    # VoxelHashMap.get_vfm_correspondences) give the same values; chaining THEM re-uploads 387-column fp64 rows at every
    # step (85 ms for a 6 000-point scan against a 30 000-point map, against 6 ms this way).
    def _correspond(self, voxel_map, raw_scan, initial_pose):
        vs = self.config.mapping.voxel_size
        # Only the scan's COORDINATES go to the device up front (N x 3): the three chained voxelisations key on them alone, and of
        # the N x 387 rows only the few hundred to ~2000 that survive the 5 m grid (RN:414) are ever searched -- their descriptor
        # columns are gathered on the host by the composed index chain and uploaded then (1 - 3 MB instead of 31 MB at C2 size).
        scan = np.asarray(raw_scan)
        if scan.ndim != 2 or scan.shape[1] < 3:
            raise ValueError("Invalid shape")
        raw_xyz = torch.from_numpy(np.ascontiguousarray(scan[:, :3], dtype=np.float64)).cuda()
        pose = np.ascontiguousarray(initial_pose, dtype=np.float64)
        key = pose.tobytes()
        if self._pose_cache is None or self._pose_cache[0] != key:
            self._pose_cache = (key, torch.from_numpy(pose).cuda())
        T = self._pose_cache[1]
        chain = self._voxel_chain(raw_xyz, T, vs) if 1 <= len(raw_xyz) <= (1 << 18) else None
        if chain is not None:
            # RN:399-400 + 414 in three launches and ONE read-back (round 5): every level takes the survivors of the one before as rows of
            # the raw scan, on the device
            xyz, raw_of_voxel_scan, order5, raw_idx5 = chain
        else:
            o1 = ops.voxel_robin(raw_xyz, vs * 0.5)                     # RN:399
            xyz = raw_xyz[o1]
            o2 = ops.voxel_robin(xyz, vs * 1.0)                         # RN:400  -> voxel_scan
            xyz = xyz[o2]
            raw_of_voxel_scan = o1[o2]                                  # rows of raw_scan behind voxel_scan, in its order
            order5 = raw_idx5 = None
        voxel_hash_map = self._hash_map_for(voxel_map)                  # RN:402-403 (kept across the scans of a scene)
        pcl_xyz = ops.transform_xyz(xyz, T)                             # RN:408 (descriptors carried through)
        out = None
        for voxel in (5.0, 1.0):                                        # RN:414, retry RN:420-423
            if voxel == 5.0 and order5 is not None:
                order, raw_idx = order5, raw_idx5
            else:
                order = ops.voxel_robin(pcl_xyz, voxel)
                raw_idx = raw_of_voxel_scan[order].cpu().numpy()
            sub_xyz = pcl_xyz[order]
            q_desc = self._upload_rows(scan, raw_idx)                   # VoxelHashMap.cpp:478-481
            qi, mi, _ = voxel_hash_map.search_device(None, self.min_cosine_similarity, q_desc=q_desc)       # RN:418
            out = dict(src_rows=order[qi], tgt_rows=mi, src_xyz=sub_xyz[qi])
            if len(qi) >= 75:
                break
            if voxel == 5.0:
                print("[WARNING] Voxelized too sparse, retrying with a larger voxel size")
        out.update(voxel_scan_xyz=xyz, voxel_hash_map=voxel_hash_map, map_xyz=voxel_hash_map.point_cloud_device())
        return out

    def _voxel_chain(self, raw_xyz: torch.Tensor, T: torch.Tensor, vs: float):
        """The three chained VoxelDownsample()s of RN:399-400 and 414 (voxel sizes vs / 2, vs, 5 m; the third on the points moved by the
        initial pose) as three kernel launches and one read-back: (voxel_scan xyz, its rows of the raw scan, the 5 m level's positions in
        voxel_scan, their rows of the raw scan on the host) -- or None where a level is not the one-launch kernel's (the caller then takes
        the level-by-level path: the same answers)."""
        n = raw_xyz.shape[0]
        l1 = ops.voxel_robin_level(raw_xyz, vs * 0.5)
        l2 = ops.voxel_robin_level(raw_xyz, vs * 1.0, idx=l1["keep"], n_dev=l1["count"], n_max=n)
        l3 = ops.voxel_robin_level(raw_xyz, 5.0, idx=l2["keep"], n_dev=l2["count"], n_max=n, T=T, want_local=True)
        head = min(n, self._CHAIN_HEAD)
        if self._chain_host is None:
            self._chain_host = torch.empty(24 + self._CHAIN_HEAD, dtype=torch.int64).pin_memory()
        packed = torch.cat((l1["info"], l2["info"], l3["info"], l3["keep"][:head]))
        self._chain_host[:packed.numel()].copy_(packed, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h = self._chain_host.numpy()
        if any(h[8 * k + 5] != 1 or h[8 * k + 1] < 0 for k in range(3)):
            return None
        n2, n3 = int(h[9]), int(h[17])
        if n2 == 0 or n3 == 0:
            return None
        raw_of_voxel_scan = l2["keep"][:n2]
        raw_idx5 = h[24:24 + n3].copy() if n3 <= head else l3["keep"][:n3].cpu().numpy()
        return raw_xyz[raw_of_voxel_scan], raw_of_voxel_scan, l3["local"][:n3], raw_idx5

    _CHAIN_HEAD = 8192   # rows of the 5 m level read back together with the levels' infos (more: a second read-back)

    @staticmethod
    def _upload_rows(scan: np.ndarray, raw_idx: np.ndarray) -> torch.Tensor:
        """scan[raw_idx, 3:] as float32 on the device.  (Measured and dropped in round 5: gathering into a page-locked staging buffer
        and an asynchronous copy from there -- 180 us against 135 for this form at 1 700 rows x 384: numpy's row gather into
        page-locked memory is slower than into pageable memory by more than the driver's staging copy costs.)"""
        return torch.from_numpy(np.ascontiguousarray(scan[raw_idx, 3:], dtype=np.float32)).cuda()

    def compute_vfm_correspondences(self, voxel_map, raw_scan, initial_pose=np.eye(4)):
        c = self._correspond(voxel_map, raw_scan, initial_pose)
        return c["src_xyz"].cpu().numpy(), c["map_xyz"][c["tgt_rows"]].cpu().numpy()

    def ransac_registration(self, voxel_map, raw_scan, method: str = "vfm", run_icp: bool = False):
        if method != "vfm":
            raise ValueError(f"Invalid method: {method}")  # baselines are out of scope
        c = self._correspond(voxel_map, raw_scan, np.eye(4))
        # correspondence indices (RN:288-317).  The reference re-voxelises the scan and the map in 3-D and recovers the
        # rows with two KD-trees (distance < 1e-3); the containers are the same ones (same hash, same order), so the rows
        # are the ones the search already produced.
        # (the clouds and the index pairs stay on the device: o3d.utility.DeviceArray -- np.asarray() of one downloads it on demand;
        # the map's 200 000 points used to go down and up again, 4.8 MB each way, in every call)
        pcd_src = o3d.geometry.PointCloud()
        pcd_src.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(c["voxel_scan_xyz"]))
        pcd_tgt = o3d.geometry.PointCloud()
        pcd_tgt.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(c["map_xyz"]))
        coors = o3d.utility.Vector2iVector(o3d.utility.DeviceArray(torch.stack((c["src_rows"], c["tgt_rows"]), dim=1).to(torch.int32)))
        result = o3d.pipelines.registration.registration_ransac_based_on_correspondence(
            pcd_src, pcd_tgt, coors, self.max_correspondence_distance,
            o3d.pipelines.registration.TransformationEstimationPointToPoint(False), ransac_n=3,
            criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(self.ransac_iterations, 1))
        ransac_pose = np.array(result.transformation)
        if run_icp:
            ransac_pose = orthogonalize_rotation(ransac_pose)                 # RN:331-336
            sigma = self.config.adaptive_threshold.initial_threshold          # RN:339
            # RN:290-293 builds the 3-D hash map from voxel_map[:, :3]: the same kept points in the same container
            vhm = c["voxel_hash_map"]
            voxel_scan = np.asarray(pcd_src.points)
            pose = register_frame(points=voxel_scan, voxel_map=vhm if not vhm.empty() else vhm.xyz_map(), initial_guess=ransac_pose,
                                  max_correspondance_distance=3 * sigma, kernel=sigma / 3)   # RN:340-344
            return ransac_pose, pose
        return ransac_pose, None

