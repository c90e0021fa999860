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


def Vector3dVector(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 3:
        raise RuntimeError("Vector3dVector: expected an N x 3 array")
    return a


def Vector2iVector(a):
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
        self.correspondence_set = np.zeros((0, 2), dtype=np.int32)

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
    src = torch.from_numpy(Vector3dVector(np.asarray(source.points))).cuda()
    tgt = torch.from_numpy(Vector3dVector(np.asarray(target.points))).cuda()
    cs = Vector2iVector(corres)
    res = RegistrationResult()
    if len(cs) < ransac_n or max_correspondence_distance <= 0.0:
        return res  # Open3D returns the default result
    if cs.min() < 0 or cs[:, 0].max() >= len(src) or cs[:, 1].max() >= len(tgt):
        raise IndexError("correspondence index out of range")
    out = ops.ransac_corr(src, tgt, torch.from_numpy(cs).cuda(), float(max_correspondence_distance),
                          criteria.max_iteration, seed=_seed[0] if seed is None else seed)
    res.transformation = out["T"].cpu().numpy()
    res.fitness = float(out["fitness"].item())
    res.inlier_rmse = float(out["rmse"].item())
    res.correspondence_set = cs[out["mask"][:len(cs)].cpu().numpy().astype(bool)]
    res.best_hypothesis = int(out["best_hyp"].item())
    return res


geometry = SimpleNamespace(PointCloud=PointCloud)
utility = SimpleNamespace(Vector3dVector=Vector3dVector, Vector2iVector=Vector2iVector,
                          random=SimpleNamespace(seed=_seed_fn))
pipelines = SimpleNamespace(registration=SimpleNamespace(
    registration_ransac_based_on_correspondence=registration_ransac_based_on_correspondence,
    TransformationEstimationPointToPoint=TransformationEstimationPointToPoint,
    RANSACConvergenceCriteria=RANSACConvergenceCriteria, RegistrationResult=RegistrationResult))
