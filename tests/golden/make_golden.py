"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE's own functions.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the reference
travels: the fixtures hold seeded INPUTS and the reference's OUTPUTS as .npz data.

    python tests/golden/make_golden.py

Importable reference callables:
  dataloader.nclt.NCLT.project_pcl_to_image                 -> proj_nclt.npz
  dataloader.oxford_robotcar.OxfordRobotcar.project_pcl_to_image -> proj_oxf.npz
  dataloader.kitti_odometry.KittiOdometry.project_pcl_to_image   -> proj_kitti.npz
  prepare_scenes.create_descriptors                         -> lift_oxf.npz, lift_nclt.npz
  vfm_reg.utils.transform_pcl                               -> transform_pcl.npz
  pointdsc.common.rigid_transform_3d                        -> kabsch_dsc.npz
  print_errors.compute_success_rate / print_errors.main     -> print_errors.npz (recall + the paper-table rows)

Libraries the reference imports at module level but that are absent here are replaced by
``MagicMock`` (they are never touched by the functions above) with one exception:
``cv2.rotate(img, cv2.ROTATE_90_COUNTERCLOCKWISE)`` (prepare_scenes.py:73-74) is given as
``np.rot90(img, 1)``, the same 90-degree counter-clockwise array rotation.
"""
import sys
import types
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn.functional as F

REF = Path("/root/reference/src/vfm-reg/src")
OUT = Path(__file__).resolve().parent


def _install_stubs():
    names = [
        "rospy", "faiss", "tf_conversions", "geometry_msgs", "geometry_msgs.msg", "sensor_msgs",
        "sensor_msgs.msg", "visualization_msgs", "visualization_msgs.msg", "colour_demosaicing",
        "h5py", "kiss_icp", "kiss_icp.voxelization", "kiss_icp.pybind", "featup",
        "featup.featurizers", "featup.featurizers.maskclip", "featup.featurizers.maskclip.clip",
        "featup.util", "pytorch_lightning", "torchvision", "torchvision.transforms", "matplotlib",
        "matplotlib.pyplot", "PIL", "PIL.Image",
    ]
    for n in names:
        if n not in sys.modules:
            try:
                __import__(n)
            except Exception:
                sys.modules[n] = MagicMock()
    cv2 = types.ModuleType("cv2")
    cv2.ROTATE_90_COUNTERCLOCKWISE = 2
    cv2.rotate = lambda img, code: np.ascontiguousarray(np.rot90(img, 1)) if code == 2 else None
    sys.modules["cv2"] = cv2
    sys.path.insert(0, str(REF))


def _pose(rng, yaw_deg, t):
    from scipy.spatial.transform import Rotation as R
    T = np.eye(4)
    T[:3, :3] = R.from_euler("xyz", [rng.normal(0, 2), rng.normal(0, 2), yaw_deg], degrees=True).as_matrix()
    T[:3, 3] = t
    return T


def _cloud(rng, n):
    xyz = np.c_[rng.uniform(-30, 30, n), rng.uniform(-30, 30, n), rng.uniform(-2, 6, n)]
    return xyz.astype(np.float32)


def gen_proj_nclt(rng):
    from dataloader.nclt import NCLT
    ds = object.__new__(NCLT)
    cam = "Cam3"
    ds.cameras = [cam]
    ds.image_subsample = 2
    K = np.array([[410.0, 0.0, 805.3], [0.0, 409.1, 612.9], [0.0, 0.0, 1.0]])
    x_lb3_c = _pose(rng, 72.0, [0.01, -0.04, 0.02])
    # Ladybug cameras look along +z with the sensor rotated: make the cloud visible
    from scipy.spatial.transform import Rotation as R
    x_lb3_c[:3, :3] = R.from_euler("xyz", [-90.0, 0.0, 72.0], degrees=True).as_matrix()
    ds.camera_parameters = {cam: {"K": K, "x_lb3": x_lb3_c}}
    coords = [210, 450, 820, 700]
    ds.undistortion_masks = {cam: {"coords": coords}}
    n = 6000
    xyz = _cloud(rng, n)
    pcl = np.insert(xyz, 3, values=1, axis=1).T  # PS:69 (float32 4xN)
    mc = np.array(coords) // ds.image_subsample
    image = np.full((mc[2], mc[3], 3), 200, dtype=np.uint8)
    image[40:90, 30:200] = 0  # planted black region (NCLT:353-359)
    image[300:310, :] = 0
    image[5, 7] = (0, 0, 1)  # not black: one channel non-zero
    u, v, idx = ds.project_pcl_to_image(pcl, image, cam)
    # what the build needs as inputs: the composed extrinsic exactly as the reference forms it
    x_body_lb3 = np.eye(4)
    x_body_lb3[:3, 3] = [0.035, 0.002, -1.23]
    x_body_lb3[:3, :3] = R.from_euler("xyz", [-179.93, -0.23, 0.50], degrees=True).as_matrix()
    T_c_body = np.linalg.inv(x_lb3_c) @ np.linalg.inv(x_body_lb3)
    np.savez_compressed(OUT / "proj_nclt.npz", pcl=pcl, K=K, x_lb3_c=x_lb3_c, T_c_body=T_c_body,
                        subsample=ds.image_subsample, coords=np.array(coords), image=image,
                        u=np.asarray(u), v=np.asarray(v), idx=np.asarray(idx))
    print("proj_nclt", len(idx), "of", n)


def gen_proj_oxf(rng):
    from dataloader.oxford_robotcar import OxfordRobotcar
    ds = object.__new__(OxfordRobotcar)
    cam = "mono_left"
    ds.cameras = [cam]
    ds.image_subsample = 2
    lidar_in_ego = _pose(rng, 3.0, [1.1, 0.0, -0.3])
    cam_in_ego = np.linalg.inv(_pose(rng, 88.0, [0.3, 0.2, -0.4]))
    G = np.array([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    cm = types.SimpleNamespace(G_camera_image=G, focal_length=(400.0, 398.5), principal_point=(508.2, 498.7))
    ds.calib = {"lidar_in_ego": lidar_in_ego, f"{cam}_in_ego": cam_in_ego}
    ds.camera_model = {cam: cm}
    n = 6000
    xyz = _cloud(rng, n)
    # plant the reference's boundary quirks: a point exactly on z == 0 and far-away points
    pcl = np.insert(xyz, 3, values=1, axis=1).T
    image = np.zeros((512, 512, 3), dtype=np.uint8)
    u, v, idx = ds.project_pcl_to_image(pcl, image, cam)
    Ginv = np.linalg.solve(G, np.eye(4))
    np.savez_compressed(OUT / "proj_oxf.npz", pcl=pcl, lidar_in_ego=lidar_in_ego, cam_in_ego=cam_in_ego,
                        G=G, Ginv=Ginv, fc=np.array([cm.focal_length[0], cm.focal_length[1],
                                                     cm.principal_point[0], cm.principal_point[1]]),
                        subsample=ds.image_subsample, H=image.shape[0], W=image.shape[1],
                        u=np.asarray(u), v=np.asarray(v), idx=np.asarray(idx))
    print("proj_oxf", len(idx), "of", n, "u max", np.max(u), "v max", np.max(v))


def gen_proj_kitti(rng):
    from dataloader.kitti_odometry import KittiOdometry
    ds = object.__new__(KittiOdometry)
    ds.image_subsample = 1
    P2 = np.array([[718.856, 0.0, 607.1928, 45.38225], [0.0, 718.856, 185.2157, -0.1130887],
                   [0.0, 0.0, 1.0, 0.003779761]])
    Tr = np.array([[4.276802385584e-04, -9.999672484946e-01, -8.084491683471e-03, -1.198459927713e-02],
                   [-7.210626507497e-03, 8.081198471645e-03, -9.999413164504e-01, -5.403984729748e-02],
                   [9.999738645903e-01, 4.859485810390e-04, -7.206933692422e-03, -2.921968648686e-01],
                   [0.0, 0.0, 0.0, 1.0]])
    ds.calib = {"P2": P2, "Tr_velo_to_cam": Tr}
    n = 6000
    xyz = _cloud(rng, n)
    pcl = np.insert(xyz, 3, values=1, axis=1).T
    image = np.zeros((376, 1241, 3), dtype=np.uint8)
    u, v, idx = ds.project_pcl_to_image(pcl, image, "camera")
    np.savez_compressed(OUT / "proj_kitti.npz", pcl=pcl, P2=P2, Tr=Tr, P2Tr=P2 @ Tr, subsample=1,
                        H=image.shape[0], W=image.shape[1], u=np.asarray(u), v=np.asarray(v),
                        idx=np.asarray(idx))
    print("proj_kitti", len(idx), "of", n)


class _FakeFeatures:
    """feature_generator duck-type: get_image_features(image, upsample=True) as IF:79-110 does
    after the backbone: bilinear upsample of a (seeded) patch grid to the image size, HWC numpy."""

    def __init__(self, grids):
        self.grids = grids
        self.calls = 0

    def get_image_features(self, image, upsample=False, cache_file=""):
        g = torch.from_numpy(self.grids[self.calls]).permute(2, 0, 1).unsqueeze(0)
        self.calls += 1
        f = F.interpolate(g, image.shape[:2], mode="bilinear", align_corners=False)
        return f.squeeze().permute(1, 2, 0).cpu().numpy()


def gen_lift_oxf(rng):
    import prepare_scenes as PS
    from dataloader.oxford_robotcar import OxfordRobotcar
    cams = ["stereo/centre", "mono_left", "mono_right"]
    seq = object.__new__(OxfordRobotcar)
    seq.cameras = cams
    seq.image_subsample = 1
    H, W, Cc = 96, 128, 8
    G = np.array([[0.0, 0.0, 1.0, 0.0], [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    seq.calib = {"lidar_in_ego": _pose(rng, 1.0, [0.5, 0.0, -0.2])}
    seq.camera_model = {}
    yaws = [0.0, 50.0, -50.0]  # overlapping fields of view -> first-camera-wins is exercised
    for c, yaw in zip(cams, yaws):
        seq.calib[f"{c}_in_ego"] = np.linalg.inv(_pose(rng, yaw, [0.2, 0.0, -0.1]))
        seq.camera_model[c] = types.SimpleNamespace(G_camera_image=G, focal_length=(60.0, 60.0),
                                                    principal_point=(64.0, 48.0))
    images = {}
    for c in cams:
        img = rng.integers(1, 255, size=(H, W, 3), dtype=np.uint8)
        img[10:30, 40:90] = 0  # black block -> zero descriptors (PS:57-62)
        images[c] = img
    seq.read_images = lambda filenames: images
    grids = [rng.standard_normal((7, 9, Cc)).astype(np.float32) for _ in cams]
    fg = _FakeFeatures(grids)
    n = 4000
    xyz = _cloud(rng, n)
    desc = PS.create_descriptors(None, seq, fg, xyz)
    np.savez_compressed(OUT / "lift_oxf.npz", xyz=xyz, images=np.stack([images[c] for c in cams]),
                        grids=np.stack(grids), lidar_in_ego=seq.calib["lidar_in_ego"],
                        cam_in_ego=np.stack([seq.calib[f"{c}_in_ego"] for c in cams]), G=G,
                        Ginv=np.linalg.solve(G, np.eye(4)),
                        fc=np.array([60.0, 60.0, 64.0, 48.0]), subsample=1, desc=desc)
    print("lift_oxf", int((np.abs(desc).sum(1) > 0).sum()), "of", n, "points lifted")


def gen_lift_nclt(rng):
    import prepare_scenes as PS
    from dataloader.nclt import NCLT
    from scipy.spatial.transform import Rotation as R
    cams = ["Cam1", "Cam2"]
    seq = object.__new__(NCLT)
    seq.cameras = cams
    seq.image_subsample = 1
    coords = [12, 20, 120, 90]  # [row0, col0, h, w] of the crop window in the ROTATED frame
    seq.undistortion_masks = {c: {"coords": coords} for c in cams}
    seq.camera_parameters = {}
    for c, yaw in zip(cams, [20.0, 75.0]):
        x = np.eye(4)
        x[:3, :3] = R.from_euler("xyz", [-90.0, 0.0, yaw], degrees=True).as_matrix()
        x[:3, 3] = [0.02, -0.01, 0.03]
        seq.camera_parameters[c] = {"K": np.array([[70.0, 0.0, 65.0], [0.0, 70.0, 72.0], [0.0, 0.0, 1.0]]),
                                    "x_lb3": x}
    Hraw, Wraw, Cc = 90, 120, 8  # raw (un-rotated) image: rotated one is 120 x 90 = window h x w
    images = {}
    for c in cams:
        img = rng.integers(1, 255, size=(Hraw, Wraw, 3), dtype=np.uint8)
        img[20:45, 30:70] = 0
        images[c] = img
    seq.read_images = lambda filenames: images
    grids = [rng.standard_normal((6, 8, Cc)).astype(np.float32) for _ in cams]
    fg = _FakeFeatures(grids)
    n = 4000
    xyz = _cloud(rng, n)
    desc = PS.create_descriptors(None, seq, fg, xyz)
    x_body_lb3 = np.eye(4)
    x_body_lb3[:3, 3] = [0.035, 0.002, -1.23]
    x_body_lb3[:3, :3] = R.from_euler("xyz", [-179.93, -0.23, 0.50], degrees=True).as_matrix()
    T_c_body = np.stack([np.linalg.inv(seq.camera_parameters[c]["x_lb3"]) @ np.linalg.inv(x_body_lb3)
                         for c in cams])
    np.savez_compressed(OUT / "lift_nclt.npz", xyz=xyz, images=np.stack([images[c] for c in cams]),
                        grids=np.stack(grids), K=np.stack([seq.camera_parameters[c]["K"] for c in cams]),
                        x_lb3=np.stack([seq.camera_parameters[c]["x_lb3"] for c in cams]),
                        T_c_body=T_c_body, coords=np.array(coords), subsample=1, desc=desc)
    print("lift_nclt", int((np.abs(desc).sum(1) > 0).sum()), "of", n, "points lifted")


def gen_transform_pcl(rng):
    from vfm_reg.utils import transform_pcl
    pcl = np.c_[_cloud(rng, 500), rng.standard_normal((500, 5)).astype(np.float32)].astype(np.float32)
    T = _pose(rng, 33.0, [4.0, -2.0, 0.5])
    out32 = transform_pcl(pcl, T)
    pcl64 = pcl.astype(np.float64)
    out64 = transform_pcl(pcl64, T)
    np.savez_compressed(OUT / "transform_pcl.npz", pcl=pcl, T=T, out32=out32, out64=out64)
    print("transform_pcl ok")


def gen_kabsch(rng):
    from pointdsc.common import rigid_transform_3d
    bs, n = 6, 40
    A = rng.uniform(-10, 10, (bs, n, 3)).astype(np.float32)
    Ts = np.stack([_pose(rng, rng.uniform(-180, 180), rng.normal(0, 5, 3)) for _ in range(bs)])
    B = (np.einsum("bij,bnj->bni", Ts[:, :3, :3], A) + Ts[:, None, :3, 3]).astype(np.float32)
    B += rng.normal(0, 0.01, B.shape).astype(np.float32)
    w = rng.uniform(0.1, 1.0, (bs, n)).astype(np.float32)
    T_unw = rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(B)).numpy()
    T_w = rigid_transform_3d(torch.from_numpy(A), torch.from_numpy(B), torch.from_numpy(w.copy())).numpy()
    # mirrored configuration -> exercises the det-sign fix (common.py:40-43)
    A3 = A[:, :3].copy()
    B3 = B[:, :3].copy()
    T_3 = rigid_transform_3d(torch.from_numpy(A3), torch.from_numpy(B3)).numpy()
    np.savez_compressed(OUT / "kabsch_dsc.npz", A=A, B=B, w=w, T_unw=T_unw, T_w=T_w, A3=A3, B3=B3, T_3=T_3,
                        T_true=Ts)
    print("kabsch ok; planted-vs-recovered max err", np.abs(T_unw - Ts).max())


def gen_print_errors():
    """print_errors.py:8-36 + the rows main() writes (print_errors.py:38-56).  main() writes `error.txt` next to the script
    (inside the read-only reference tree): the write is redirected to a temporary file, nothing else is changed."""
    import builtins
    import pickle
    import tempfile
    import print_errors as PE
    rng = np.random.default_rng(77)
    methods = ["fpfh_ransac", "fpfh_ransac_icp", "vfm_ransac", "vfm_ransac_icp", "vfm_teaser", "vfm_teaser_icp", "icp"]
    rot = {m: np.abs(rng.standard_cauchy(23)) * (0.4 if "vfm" in m else 3.0) for m in methods}
    trans = {m: np.abs(rng.standard_cauchy(23)) * (0.2 if "vfm" in m else 1.5) for m in methods}
    thresholds = [(.3, 15), (.6, 1.5), (2, 5)]
    rates = np.array([[PE.compute_success_rate(trans[m], rot[m], *t) for t in thresholds] for m in methods])
    with tempfile.TemporaryDirectory() as td:
        src = Path(td) / "errors.pkl"
        with open(src, "wb") as f:
            pickle.dump({"rot": {k: list(v) for k, v in rot.items()}, "trans": {k: list(v) for k, v in trans.items()}}, f)
        real_open = builtins.open
        out = Path(td) / "error.txt"

        def redirected(file, *a, **k):
            return real_open(out if str(file).endswith("error.txt") else file, *a, **k)
        builtins.open = redirected
        try:
            PE.main(src)
        finally:
            builtins.open = real_open
        text = out.read_text()
    np.savez(OUT / "print_errors.npz", methods=np.array(methods), rot=np.stack([rot[m] for m in methods]),
             trans=np.stack([trans[m] for m in methods]), thresholds=np.array(thresholds), rates=rates, error_txt=np.array(text))
    print("print_errors.npz", rates.shape, len(text))


def main():
    _install_stubs()
    rng = np.random.default_rng(20250620)
    gen_proj_nclt(rng)
    gen_proj_oxf(rng)
    gen_proj_kitti(rng)
    gen_lift_oxf(rng)
    gen_lift_nclt(rng)
    gen_transform_pcl(rng)
    gen_kabsch(rng)
    gen_print_errors()


if __name__ == "__main__":
    main()
