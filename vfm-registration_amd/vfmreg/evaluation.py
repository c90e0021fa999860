"""Row F4: the evaluation harness around the accelerated path -- map accumulation, error and recall
bookkeeping of ``RegistrationNode.make_step`` (registration_node.py:548-989), ``compute_errors`` /
``compute_success_rate`` (RN:997-1025) and the summary of print_errors.py:8-36.  Pure host
orchestration (as in the reference); the heavy steps it calls (voxel_down_sample, transform_pcl,
ransac_registration, register_frame) run on the GPU.

Scenes: the reference stores processed scenes as HDF5 (prepare_scenes.py:16-47, read_h5.py:17-49);
h5py is not available in this environment, so ``save_scene`` / ``read_scenes`` keep the same logical
layout (``map/<seq>/pose/<jjj>``, ``map/<seq>/point_cloud/<jjj>``, ``scans/<seq>/{pose,point_cloud}``,
rows = [x, y, z, d0..dC-1] fp32) in a ``.npz`` container (row F3 stand-in, documented in DESIGN.md).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from .registration import RegistrationNode
from .utils import transform_pcl
from .voxelization import voxel_down_sample

SUCCESS_THRESHOLDS = [(.3, 15), (.6, 1.5), (2, 5)]  # (RTE m, RRE deg): PointDSC, GCL, D3Feat/SpinNet (RN:973-977)


def save_scene(filename, sequences: List[str], map_poses, map_point_clouds, seq_poses, seq_point_clouds) -> None:
    """prepare_scenes.save_scene (PS:16-47) with the HDF5 group paths as npz keys."""
    filename = Path(filename)
    filename.parent.mkdir(parents=True, exist_ok=True)
    data = {}
    for j in range(len(map_poses)):
        data[f"map/{sequences[0]}/pose/{j:03}"] = np.asarray(map_poses[j])
        data[f"map/{sequences[0]}/point_cloud/{j:03}"] = np.asarray(map_point_clouds[j])
    for j in range(len(seq_poses)):
        if seq_poses[j] is None:  # this sequence has no hits (PS:41-42)
            continue
        data[f"scans/{sequences[j + 1]}/pose"] = np.asarray(seq_poses[j])
        data[f"scans/{sequences[j + 1]}/point_cloud"] = np.asarray(seq_point_clouds[j])
    np.savez(filename, **data)


def read_scenes(filename) -> Dict[str, list]:
    """read_h5.read_scenes (read_h5.py:17-49): dict(map_poses, map_point_clouds, scene_poses,
    scene_point_clouds, scene_sequences)."""
    z = np.load(filename)
    keys = sorted(z.files)
    map_ids = sorted({k.split("/")[3] for k in keys if k.startswith("map/") and "/pose/" in k})
    seq0 = next(k.split("/")[1] for k in keys if k.startswith("map/"))
    scans = sorted({k.split("/")[1] for k in keys if k.startswith("scans/")})
    return {
        "map_poses": [z[f"map/{seq0}/pose/{j}"] for j in map_ids],
        "map_point_clouds": [z[f"map/{seq0}/point_cloud/{j}"] for j in map_ids],
        "scene_poses": [z[f"scans/{s}/pose"] for s in scans],
        "scene_point_clouds": [z[f"scans/{s}/point_cloud"] for s in scans],
        "scene_sequences": scans,
    }


def build_local_map(map_poses, map_point_clouds, voxel_size: float = .25, n_descriptors: int = 384) -> np.ndarray:
    """RN:556-580: drop rows without descriptors (sum <= 0, RN:562), voxelise each cloud, move it into
    the map frame, concatenate, voxelise again (split in two halves above 1e6 points, RN:568-578)."""
    local = []
    for pose, pcl in zip(map_poses, map_point_clouds):
        pcl = pcl[np.sum(pcl[:, 3:], axis=1) > 0]
        pcl = voxel_down_sample(pcl, voxel_size).astype(pcl.dtype)
        local.append(transform_pcl(pcl, pose))
    m = np.concatenate(local, axis=0).astype(np.float32)
    if m.shape[0] > 1000000:
        mean_3d = np.mean(m[:, :3], axis=0)
        a = voxel_down_sample(m[m[:, 0] > mean_3d[0]], voxel_size).astype(m.dtype)
        b = voxel_down_sample(m[m[:, 0] <= mean_3d[0]], voxel_size).astype(m.dtype)
        m = np.concatenate([a, b], axis=0)
    else:
        m = voxel_down_sample(m, voxel_size).astype(m.dtype)
    return m[:, :3 + n_descriptors]


class Evaluation:
    """Error lists per method + the recall table (RN:86-89, 961-1025)."""

    def __init__(self):
        self.rot_errors: Dict[str, List[float]] = {}
        self.trans_errors: Dict[str, List[float]] = {}
        self.points_in_map: List[int] = []

    def compute_errors(self, pose: np.ndarray, gt_pose: np.ndarray, method: str) -> Tuple[float, float]:
        R, R_gt = pose[:3, :3], gt_pose[:3, :3]
        rot_error = abs(np.arccos(min(max(((R.T @ R_gt).trace() - 1) / 2, -1.0), 1.0)))
        rot_error = float(np.rad2deg(rot_error))
        trans_error = float(np.linalg.norm(pose[:3, 3] - gt_pose[:3, 3]))
        self.rot_errors.setdefault(method, []).append(rot_error)
        self.trans_errors.setdefault(method, []).append(trans_error)
        return trans_error, rot_error

    def compute_success_rate(self, method: str, translation_threshold, rotation_threshold) -> float:
        ok_t = np.array(self.trans_errors[method]) < translation_threshold
        ok_r = np.array(self.rot_errors[method]) < rotation_threshold
        return float(np.mean(ok_t & ok_r))

    def summary(self) -> str:
        lines = ["=" * 80]
        for method, e in self.rot_errors.items():
            lines.append(f"Rotation error ({method:<20}): {np.mean(e):.3f} ± {np.std(e):.3f}")
        lines.append("-" * 80)
        for method, e in self.trans_errors.items():
            lines.append(f"Translat error ({method:<20}): {np.mean(e):.3f} ± {np.std(e):.3f}")
        lines.append("-" * 80)
        head = f"{'':<20}: " + "".join(f"{t[0]:>3}, {t[1]:<3} | " for t in SUCCESS_THRESHOLDS)
        lines.append(head[:-2])
        for method in self.rot_errors:
            row = f"{method:<20}: " + "".join(f"{100 * self.compute_success_rate(method, *t):>8.2f} | "
                                              for t in SUCCESS_THRESHOLDS)
            lines.append(row[:-2])
        lines.append("=" * 80)
        return "\n".join(lines)


def evaluate_scene(scene: Dict[str, list], node: Optional[RegistrationNode] = None,
                   evaluation: Optional[Evaluation] = None, run_icp: bool = True) -> Evaluation:
    """The VFM + RANSAC (+ ICP) branch of make_step for one scene (RN:587-589, 593, 860-882, 943-951):
    every scan is registered against the accumulated map with the identity as initial guess and
    compared with its ground-truth pose."""
    node = node or RegistrationNode()
    ev = evaluation or Evaluation()
    n_desc = scene["map_point_clouds"][0].shape[1] - 3
    local_map = build_local_map(scene["map_poses"], scene["map_point_clouds"], n_descriptors=n_desc)
    for gt_pose, point_cloud in zip(scene["scene_poses"], scene["scene_point_clouds"]):
        point_cloud = voxel_down_sample(point_cloud, .1).astype(point_cloud.dtype)   # RN:593
        initial_pose = np.eye(4)                                                     # RN:858
        point_cloud = transform_pcl(point_cloud, initial_pose)                       # RN:863
        results = {}
        results["vfm_ransac"], results["vfm_ransac_icp"] = node.ransac_registration(local_map, point_cloud, "vfm", run_icp)
        for k, v in results.items():
            if v is None:
                continue
            ev.compute_errors(np.asarray(gt_pose), v @ initial_pose, k)              # RN:943-947
        ev.points_in_map.append(local_map.shape[0])
    return ev
