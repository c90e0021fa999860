"""Duck-typed stand-ins for the slice of Open3D's Python API the registration path uses
(registration_node.py:312-328): ``o3d.geometry.PointCloud``, ``o3d.utility.Vector3dVector`` /
``Vector2iVector``, ``o3d.utility.random.seed`` and
``o3d.pipelines.registration.registration_ransac_based_on_correspondence`` with
``TransformationEstimationPointToPoint`` and ``RANSACConvergenceCriteria``.  Usage:

    from vfmreg import o3d            # instead of: import open3d as o3d

RANSAC runs on the GPU (csrc/ransac.hip).  Deviations from Open3D 0.18, all documented in
DESIGN.md: hypotheses are drawn from Philox4x32-10 keyed by the global seed (Open3D's mt19937 is
shared across OpenMP threads, hence not reproducible); exact ties go to the earliest hypothesis;
degenerate (collinear / repeated) samples are skipped.  Unsupported options raise.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops

_seed = [42]  # registration_node.py:39 seeds Open3D with 42


def _seed_fn(s: int):
    _seed[0] = int(s)


class PointCloud:
    def __init__(self, points=None):
        self.points = np.zeros((0, 3)) if points is None else points


class DeviceArray:
    """An array that already lives on the GPU (N x 3 fp64 points / C x 2 int32 correspondences), accepted wherever the stand-ins
    take a numpy array.  ``np.asarray()`` downloads it once, on demand; ``registration_ransac_based_on_correspondence`` reads the
    device tensor directly -- the reference-shaped call (vfmreg/registration.py) hands the map's 200 000 points over this way instead
    of downloading 4.8 MB and uploading them again in every call."""

    def __init__(self, t: torch.Tensor):
        self.device_tensor = t
        self._host = None

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            self._host = self.device_tensor.cpu().numpy()
        return self._host if dtype is None else self._host.astype(dtype, copy=False)

    def __len__(self):
        return int(self.device_tensor.shape[0])

    @property
    def shape(self):
        return tuple(self.device_tensor.shape)


def Vector3dVector(a):
    if isinstance(a, DeviceArray):
        if a.device_tensor.dim() != 2 or a.device_tensor.shape[1] != 3 or a.device_tensor.dtype != torch.float64:
            raise RuntimeError("Vector3dVector: expected an N x 3 float64 array")
        return a
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise RuntimeError("Vector3dVector: expected an N x 3 array")
    return a


def Vector2iVector(a):
    if isinstance(a, DeviceArray):
        if a.device_tensor.dim() != 2 or a.device_tensor.shape[1] != 2 or a.device_tensor.dtype != torch.int32:
            raise RuntimeError("Vector2iVector: expected a C x 2 int32 array")
        return a
    a = np.ascontiguousarray(np.asarray(a).reshape(-1, 2), dtype=np.int32)
    return a


class TransformationEstimationPointToPoint:
    def __init__(self, with_scaling: bool = False):
        self.with_scaling = bool(with_scaling)


class RANSACConvergenceCriteria:
    def __init__(self, max_iteration: int = 100000, confidence: float = 0.999):
        self.max_iteration = int(max_iteration)
        self.confidence = float(confidence)


class RegistrationResult:
    def __init__(self):
        self.transformation = np.eye(4)
        self.fitness = 0.0
        self.inlier_rmse = 0.0
        self._cs = np.zeros((0, 2), dtype=np.int32)
        self._cs_lazy = None   # (correspondences, device mask): the inlier set is gathered when somebody reads it (RN:328 reads the pose only)

    @property
    def correspondence_set(self):
        if self._cs_lazy is not None:
            cs, mask = self._cs_lazy
            cs = np.asarray(cs)
            self._cs = cs[mask[:len(cs)].cpu().numpy().astype(bool)]
            self._cs_lazy = None
        return self._cs

    @correspondence_set.setter
    def correspondence_set(self, v):
        self._cs, self._cs_lazy = v, None

    def __repr__(self):
        return (f"RegistrationResult with fitness={self.fitness:e}, inlier_rmse={self.inlier_rmse:e}, "
                f"and correspondence_set size of {len(self.correspondence_set)}")


def registration_ransac_based_on_correspondence(source, target, corres, max_correspondence_distance,
                                                estimation_method=None, ransac_n=3, checkers=(),
                                                criteria=None, seed=None) -> RegistrationResult:
    estimation_method = estimation_method or TransformationEstimationPointToPoint(False)
    criteria = criteria or RANSACConvergenceCriteria()
    if getattr(estimation_method, "with_scaling", False):
        raise NotImplementedError("with_scaling=True is outside the reference's call (registration_node.py:324)")
    if ransac_n != 3:
        raise NotImplementedError("ransac_n must be 3 (registration_node.py:325)")
    if len(checkers):
        raise NotImplementedError("correspondence checkers are not used by the reference call")
    if criteria.confidence != 1.0:
        raise NotImplementedError("confidence must be 1 (no early exit), as at registration_node.py:326")
    def points_of(pc):
        pts = pc.points
        if isinstance(pts, DeviceArray):
            return Vector3dVector(pts).device_tensor.contiguous()
        return torch.from_numpy(Vector3dVector(np.asarray(pts))).cuda()
    src, tgt = points_of(source), points_of(target)
    cs = Vector2iVector(corres)
    res = RegistrationResult()
    if len(cs) < ransac_n or max_correspondence_distance <= 0.0:
        return res  # Open3D returns the default result
    on_device = isinstance(cs, DeviceArray)
    if on_device:
        # device-resident indices: the range check Open3D makes on the host is made by the kernel that gathers the point pairs
        # (vfm_ransac_corr_bounded) and comes back with the pose -- a min / max pass and a read-back of its own were 50 us per call
        cs_dev = cs.device_tensor.contiguous()
    else:
        cs_dev = torch.from_numpy(cs).cuda()
        if cs.min() < 0 or cs[:, 0].max() >= len(src) or cs[:, 1].max() >= len(tgt):
            raise IndexError("correspondence index out of range")
    out = ops.ransac_corr(src, tgt, cs_dev, float(max_correspondence_distance),
                          criteria.max_iteration, seed=_seed[0] if seed is None else seed, check_bounds=on_device)
    # one read-back for pose, fitness, rmse, winner (and the range flag): four .item() / .cpu() calls were four synchronisations
    parts = [out["T"].reshape(-1), out["fitness"].reshape(-1), out["rmse"].reshape(-1), out["best_hyp"].reshape(-1).double()]
    if on_device:
        parts.append(out["bad"].double())
    packed = torch.cat(parts).cpu().numpy()
    if on_device and packed[19] != 0.0:
        raise IndexError("correspondence index out of range")
    res.transformation = packed[:16].reshape(4, 4).copy()
    res.fitness = float(packed[16])
    res.inlier_rmse = float(packed[17])
    res.best_hypothesis = int(packed[18])
    res._cs_lazy = (cs, out["mask"])
    return res


geometry = SimpleNamespace(PointCloud=PointCloud)
utility = SimpleNamespace(Vector3dVector=Vector3dVector, Vector2iVector=Vector2iVector, DeviceArray=DeviceArray,
                          random=SimpleNamespace(seed=_seed_fn))
pipelines = SimpleNamespace(registration=SimpleNamespace(
    registration_ransac_based_on_correspondence=registration_ransac_based_on_correspondence,
    TransformationEstimationPointToPoint=TransformationEstimationPointToPoint,
    RANSACConvergenceCriteria=RANSACConvergenceCriteria, RegistrationResult=RegistrationResult))
