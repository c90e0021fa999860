"""Mirror of the projection half of the reference's dataset classes
(src/vfm-reg/src/dataloader/{nclt,oxford_robotcar,kitti_odometry}.py):
``Dataset.project_pcl_to_image(pcl[4,N], image[H,W,3], camera) -> (u, v, pcl_indices)``.

The file readers, undistortion, demosaicing and timestamp handling of those classes are dataset
I/O and out of scope: the calibration they load is passed in.  The
projection itself runs on the GPU (csrc/project.hip) in fp64 with the reference's operation order;
integer outputs equal the reference's on the golden fixtures.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops


def _to_dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


class _Base:
    cameras: List[str]
    image_subsample: int

    def _finish(self, u, v, idx, cnt):
        k = int(cnt.item())
        return (u[:k].cpu().numpy().astype(np.int64), v[:k].cpu().numpy().astype(np.int64),
                idx[:k].cpu().numpy())

    def read_images(self, filenames=None) -> Dict[str, np.ndarray]:
        raise NotImplementedError("dataset I/O is out of scope; pass images in (see create_descriptors)")


class NCLT(_Base):
    """nclt.py: cameras Cam1..Cam5 (nclt.py:38-39); body->Ladybug extrinsic of the dataset SDK
    (nclt.py:319-321); crop window ``undistortion_masks[cam]['coords']`` = [row0, col0, h, w]."""
    X_BODY_LB3_T = [0.035, 0.002, -1.23]
    X_BODY_LB3_RPY_DEG = [-179.93, -0.23, 0.50]

    def __init__(self, camera_parameters: Dict[str, dict], undistortion_masks: Dict[str, dict], image_subsample: int = 1,
                 cameras: Optional[List[str]] = None):
        self.camera_parameters = camera_parameters
        self.undistortion_masks = undistortion_masks
        self.image_subsample = image_subsample
        self.cameras = cameras or list(camera_parameters.keys())

    def extrinsic(self, camera: str) -> np.ndarray:
        from scipy.spatial.transform import Rotation as R
        x_body_lb3 = np.eye(4)
        x_body_lb3[:3, 3] = self.X_BODY_LB3_T
        x_body_lb3[:3, :3] = R.from_euler("xyz", self.X_BODY_LB3_RPY_DEG, degrees=True).as_matrix()
        T_lb3_body = np.linalg.inv(x_body_lb3)
        T_c_lb3 = np.linalg.inv(self.camera_parameters[camera]["x_lb3"])
        return T_c_lb3 @ T_lb3_body  # nclt.py:323-325

    def projection_params(self, camera: str, image_shape) -> dict:
        """what project_pcl_to_image feeds the projection kernel (also used by the fused multi-camera lift)"""
        assert camera in self.cameras, f"Camera {camera} not available"
        K = np.asarray(self.camera_parameters[camera]["K"], dtype=np.float64)
        win = np.array(self.undistortion_masks[camera]["coords"]) // self.image_subsample
        return dict(mode=ops.PROJ_NCLT, mats=[self.extrinsic(camera), K], fc=None, subsample=float(self.image_subsample),
                    win=win, H=image_shape[0], W=image_shape[1], needs_image=True)

    def project_pcl_to_image(self, pcl, image, camera: str, _device_inputs=None):
        q = self.projection_params(camera, image.shape)
        pcl_d = _device_inputs[0] if _device_inputs else _to_dev(pcl, np.float64)
        img_d = _device_inputs[1] if _device_inputs else _to_dev(image, np.uint8)
        r = ops.project_pinhole(q["mode"], pcl_d, q["mats"], None, q["subsample"], q["win"], img_d)
        return r if _device_inputs else self._finish(*r)


class OxfordRobotcar(_Base):
    """oxford_robotcar.py: cameras stereo/centre, mono_left, mono_right, mono_rear (:35-37);
    ``calib['lidar_in_ego']``, ``calib[f'{cam}_in_ego']``; ``camera_model[cam]`` with
    G_camera_image, focal_length, principal_point (robotcar_sdk CameraModel)."""

    def __init__(self, calib: Dict[str, np.ndarray], camera_model: Dict[str, object], image_subsample: int = 1,
                 cameras: Optional[List[str]] = None):
        self.calib = calib
        self.camera_model = camera_model
        self.image_subsample = image_subsample
        self.cameras = cameras or list(camera_model.keys())

    def projection_params(self, camera: str, image_shape) -> dict:
        assert camera in self.cameras, f"Camera {camera} not available"
        cm = self.camera_model[camera]
        G = np.asarray(cm.G_camera_image, dtype=np.float64)
        Ginv = np.linalg.solve(G, np.eye(4))  # the reference solves per call (oxford_robotcar.py:341)
        fc = [cm.focal_length[0], cm.focal_length[1], cm.principal_point[0], cm.principal_point[1]]
        return dict(mode=ops.PROJ_ROBOTCAR, mats=[self.calib["lidar_in_ego"], self.calib[f"{camera}_in_ego"], Ginv], fc=fc,
                    subsample=float(self.image_subsample), win=None, H=image_shape[0], W=image_shape[1], needs_image=False)

    def project_pcl_to_image(self, pcl, image, camera: str, _device_inputs=None):
        q = self.projection_params(camera, image.shape)
        pcl_d = _device_inputs[0] if _device_inputs else _to_dev(pcl, np.float64)
        r = ops.project_pinhole(q["mode"], pcl_d, q["mats"], q["fc"], q["subsample"], None, None, q["H"], q["W"])
        return r if _device_inputs else self._finish(*r)


class KittiOdometry(_Base):
    """kitti_odometry.py: calib['P2'] (3x4), calib['Tr_velo_to_cam'] (4x4); single camera."""

    def __init__(self, calib: Dict[str, np.ndarray], image_subsample: int = 1):
        self.calib = calib
        self.image_subsample = image_subsample
        self.cameras = ["camera"]

    def projection_params(self, camera: str, image_shape) -> dict:
        P = np.asarray(self.calib["P2"], dtype=np.float64) @ np.asarray(self.calib["Tr_velo_to_cam"], dtype=np.float64)
        return dict(mode=ops.PROJ_KITTI, mats=[P], fc=None, subsample=float(self.image_subsample), win=None,
                    H=image_shape[0], W=image_shape[1], needs_image=False)

    def project_pcl_to_image(self, pcl, image, camera: str = "camera", _device_inputs=None):
        q = self.projection_params(camera, image.shape)
        pcl_d = _device_inputs[0] if _device_inputs else _to_dev(pcl, np.float64)
        r = ops.project_pinhole(q["mode"], pcl_d, q["mats"], None, q["subsample"], None, None, q["H"], q["W"])
        return r if _device_inputs else self._finish(*r)
