"""Row F4: the evaluation harness around the accelerated path -- map accumulation, error and recall
bookkeeping of ``RegistrationNode.make_step`` (registration_node.py:548-989), ``compute_errors`` /
``compute_success_rate`` (RN:997-1025) and the summary of print_errors.py:8-36.  Pure host
orchestration (as in the reference); the heavy steps it calls (voxel_down_sample, transform_pcl,
ransac_registration, register_frame) run on the GPU.

Scenes (row F3): the reference stores processed scenes as HDF5 (prepare_scenes.py:16-47, read_h5.py:17-49):
``/map/<seq>/pose/<jjj>`` f64[4,4], ``/map/<seq>/point_cloud/<jjj>`` f32[n, 3+C], ``/scans/<seq>/{pose, point_cloud}``,
rows = [x, y, z, d0..dC-1].  ``save_scene`` / ``read_scenes`` write and read that file format itself (``vfmreg.h5lite``:
h5py is not installed; the subset of HDF5 those h5py calls produce, verified against libhdf5's own tools and files);
a path ending in ``.npz`` selects a numpy container with the same key paths instead (cache format of round 1).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import h5lite
from .registration import RegistrationNode
from .utils import transform_pcl
from .voxelization import voxel_down_sample

SUCCESS_THRESHOLDS = [(.3, 15), (.6, 1.5), (2, 5)]  # (RTE m, RRE deg): PointDSC, GCL, D3Feat/SpinNet (RN:973-977)


def save_scene(filename, sequences: List[str], map_poses, map_point_clouds, seq_poses, seq_point_clouds) -> None:
    """prepare_scenes.save_scene (PS:16-47): the same groups and datasets, as an HDF5 file."""
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
    if filename.suffix == ".npz":
        np.savez(filename, **data)
        return
    tree: dict = {"map": {sequences[0]: {"pose": {}, "point_cloud": {}}}, "scans": {}}   # PS:30-38: groups exist even when empty
    for key, arr in data.items():
        node = tree
        parts = key.split("/")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = arr
    h5lite.write_h5(filename, tree)


def read_scenes(filename) -> Dict[str, list]:
    """read_h5.read_scenes (read_h5.py:17-49): dict(map_poses, map_point_clouds, map_clip, scene_poses,
    scene_point_clouds) + scene_sequences (the scan group names, in the order the arrays are listed)."""
    filename = Path(filename)
    if filename.suffix == ".npz":
        z = np.load(filename)
        tree: dict = {}
        for key in sorted(z.files):
            node = tree
            parts = key.split("/")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = z[key]
    else:
        tree = h5lite.read_h5(filename)
    map_poses, map_point_clouds, map_clip = [], [], []
    for key in tree["map"]:  # there should be only one key corresponding to the sequence name (read_h5.py:22-24)
        g = tree["map"][key]
        for pose, cloud in zip(g["pose"].values(), g["point_cloud"].values()):
            map_poses.append(pose)
            map_point_clouds.append(cloud)
        if "clip" in g:
            map_clip.extend(g["clip"].values())
    scans = tree.get("scans", {})
    return {
        "map_poses": map_poses,
        "map_point_clouds": map_point_clouds,
        "map_clip": map_clip,
        "scene_poses": [scans[s]["pose"] for s in scans],
        "scene_point_clouds": [scans[s]["point_cloud"] for s in scans],
        "scene_sequences": list(scans),
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

    def error_string(self) -> str:
        """The rows print_errors.main writes to error.txt (print_errors.py:27-56): per method
        ``RTE mean+-std & RRE mean+-std & recall & recall after ICP`` at the (0.6 m, 1.5 deg) threshold; ``*_icp`` rows
        are listed only for the vfm methods."""
        rot = {k: np.array(v) for k, v in self.rot_errors.items()}
        trans = {k: np.array(v) for k, v in self.trans_errors.items()}
        success = {m: np.logical_and(trans[m] < .6, rot[m] < 1.5) for m in rot}
        out = ""
        for method, rot_error in rot.items():
            if 'icp' in method and 'vfm' not in method:
                continue
            trans_error = trans[method]
            recall = success[method]
            out += f"{method}\t{np.round(np.mean(trans_error), 2):.2f}$\\pm${np.round(np.std(trans_error), 2):.2f}"
            out += f" & {np.round(np.mean(rot_error), 2):.2f}$\\pm${np.round(np.std(rot_error), 2):.2f}"
            out += f" & {np.round(np.mean(recall) * 100, 2):.2f}"
            recall = success.get(f"{method}_icp", recall)
            out += f" & {np.round(np.mean(recall) * 100, 2):.2f}"
            out += "\n"
        return out

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
    node = node or RegistrationNode(cache_map=True)   # local_map below is built here and never edited: one map per scene (RN:556-589)
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
