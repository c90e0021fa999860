"""GPU tests of the drop-in Python API: the reference's names and call patterns (mapping.py,
registration_node.py, prepare_scenes.py, image_features.py) on the HIP path, checked against the
oracle and the reference-generated golden fixtures."""
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as _orc
    return _orc


def _scene(n_scan=3000, n_map=20000, d=384, seed=3):
    """map / scan clouds as registration_node.py holds them: [x, y, z, d0..d383] fp32-ish rows"""
    from vfmreg import synth
    p = synth.make_pair(n_scan, n_map, d, seed=seed)
    voxel_map = np.c_[p["b_xyz"], p["b_desc"]]
    raw_scan = np.c_[p["q_xyz"], p["q_desc"]]
    return voxel_map, raw_scan, p


def test_get_vfm_correspondences_like_the_reference(orc):
    from vfmreg.config import load_config
    from vfmreg.mapping import VoxelHashMap, get_voxel_hash_map
    VoxelHashMap.quiet = True
    voxel_map, raw_scan, p = _scene()
    voxel_hash_map = get_voxel_hash_map(load_config(None, None))      # RN:402-403
    voxel_hash_map.add_points(voxel_map)
    src, tgt = voxel_hash_map.get_vfm_correspondences(raw_scan, .8)   # RN:418
    m = voxel_hash_map.point_cloud_n()
    s_ref, t_ref, qi, mi, _ = orc.get_vfm_correspondences(raw_scan, m, 0.8)
    np.testing.assert_array_equal(src, s_ref)
    np.testing.assert_array_equal(tgt, t_ref)
    assert src.dtype == np.float64 and src.shape[1] == 3 and len(src) > 1000
    q2, m2, _ = voxel_hash_map.get_vfm_correspondence_indices(raw_scan, .8)
    np.testing.assert_array_equal(q2, qi)
    np.testing.assert_array_equal(m2, mi)
    with pytest.raises(RuntimeError):          # wrong width -> py::cast_error in the reference
        voxel_hash_map.get_vfm_correspondences(raw_scan[:, :100], .8)
    with pytest.raises(ValueError, match="Invalid shape"):
        voxel_hash_map.add_points(np.zeros((5, 2)))


def test_descriptor_width_outside_fast_path_uses_exact_kernel(orc):
    from vfmreg.mapping import VoxelHashMap
    VoxelHashMap.quiet = True
    rng = np.random.default_rng(0)
    d = 32  # baseline-descriptor width: not a multiple of 128
    m = np.c_[rng.uniform(-50, 50, (800, 3)), rng.standard_normal((800, d))]
    q = np.c_[rng.uniform(-50, 50, (200, 3)), m[rng.integers(0, 800, 200), 3:] + 0.05 * rng.standard_normal((200, d))]
    vm = VoxelHashMap(1.0, 100.0, 20)
    vm.add_points(m)
    src, tgt = vm.get_vfm_correspondences(q, .8)
    s_ref, t_ref, *_ = orc.get_vfm_correspondences(q, vm.point_cloud_n(), 0.8, bruteforce=True)
    np.testing.assert_array_equal(src, s_ref)
    np.testing.assert_array_equal(tgt, t_ref)


def test_open3d_standin_ransac(orc):
    from vfmreg import o3d
    rng = np.random.default_rng(4)
    from vfmreg import synth
    T = synth.random_pose(rng)
    src = rng.uniform(-40, 40, (900, 3))
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.02, src.shape)
    corr = np.stack([np.arange(900), np.arange(900)], 1)
    corr[::3, 1] = rng.integers(0, 900, 300)   # one third wrong
    o3d.utility.random.seed(42)
    pcd_src = o3d.geometry.PointCloud()
    pcd_src.points = o3d.utility.Vector3dVector(src)
    pcd_tgt = o3d.geometry.PointCloud()
    pcd_tgt.points = o3d.utility.Vector3dVector(tgt)
    result = o3d.pipelines.registration.registration_ransac_based_on_correspondence(
        pcd_src, pcd_tgt, o3d.utility.Vector2iVector(corr), 0.5,
        o3d.pipelines.registration.TransformationEstimationPointToPoint(False), ransac_n=3,
        criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(4000, 1))
    ref = orc.ransac_corr(src, tgt, corr, 0.5, 4000, seed=42)
    np.testing.assert_array_equal(np.array(result.transformation), ref.transformation)
    assert result.fitness == ref.fitness and result.inlier_rmse == ref.inlier_rmse
    np.testing.assert_array_equal(result.correspondence_set, corr[ref.inlier_mask.astype(bool)].astype(np.int32))
    assert np.linalg.norm(result.transformation - T) < 0.05
    # Open3D's default result for too few correspondences; unsupported options are loud
    r0 = o3d.pipelines.registration.registration_ransac_based_on_correspondence(
        pcd_src, pcd_tgt, corr[:2], 0.5, criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(10, 1))
    np.testing.assert_array_equal(r0.transformation, np.eye(4))
    with pytest.raises(NotImplementedError):
        o3d.pipelines.registration.registration_ransac_based_on_correspondence(
            pcd_src, pcd_tgt, corr, 0.5, criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(10, 0.999))
    # device-resident clouds and indices (what RegistrationNode hands over): the same result, and the index check Open3D makes on the
    # host is made by the kernel that gathers the point pairs (vfm_ransac_corr_bounded) -- an index outside its cloud raises here too
    import torch
    dsrc, dtgt = o3d.geometry.PointCloud(), o3d.geometry.PointCloud()
    dsrc.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(torch.from_numpy(src).cuda()))
    dtgt.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(torch.from_numpy(tgt).cuda()))
    dcorr = torch.from_numpy(corr.astype(np.int32)).cuda()
    o3d.utility.random.seed(42)
    rd = o3d.pipelines.registration.registration_ransac_based_on_correspondence(
        dsrc, dtgt, o3d.utility.Vector2iVector(o3d.utility.DeviceArray(dcorr)), 0.5,
        criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(4000, 1))
    np.testing.assert_array_equal(np.array(rd.transformation), ref.transformation)
    assert rd.fitness == ref.fitness and rd.inlier_rmse == ref.inlier_rmse
    for row, col, val in ((5, 0, 900), (899, 1, -1), (17, 1, 900)):
        bad = dcorr.clone()
        bad[row, col] = val
        with pytest.raises(IndexError):
            o3d.pipelines.registration.registration_ransac_based_on_correspondence(
                dsrc, dtgt, o3d.utility.Vector2iVector(o3d.utility.DeviceArray(bad)), 0.5,
                criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(100, 1))
        hb = corr.copy()
        hb[row, col] = val
        with pytest.raises(IndexError):
            o3d.pipelines.registration.registration_ransac_based_on_correspondence(
                pcd_src, pcd_tgt, o3d.utility.Vector2iVector(hb), 0.5, criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(100, 1))


def test_ransac_registration_end_to_end(orc):
    """RegistrationNode.ransac_registration(voxel_map, raw_scan, 'vfm') (RN:273-328) against the same
    steps assembled from the oracle."""
    from vfmreg import o3d
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import RegistrationNode, compute_errors, orthogonalize_rotation
    VoxelHashMap.quiet = True
    voxel_map, raw_scan, p = _scene(n_scan=6000, n_map=30000, seed=11)
    node = RegistrationNode(ransac_iterations=5000)
    o3d.utility.random.seed(42)
    pose, pose_icp = node.ransac_registration(voxel_map, raw_scan, "vfm")
    assert pose_icp is None
    # oracle re-enactment of RN:396-425 + 288-327
    vs = node.config.mapping.voxel_size
    scan = orc.voxel_down_sample(orc.voxel_down_sample(raw_scan, vs * 0.5), vs * 1.0)   # RN:399-400, container order
    mp = voxel_map[orc.voxel_hash_map_points(voxel_map, vs, 20)]                         # RN:402-403 -> PointcloudN()
    sub = orc.voxel_down_sample(scan, 5.0)                                               # RN:414
    _, _, qi, mi, _ = orc.get_vfm_correspondences(sub, mp, 0.8)
    if len(qi) < 75:
        sub = orc.voxel_down_sample(scan, 1.0)
        _, _, qi, mi, _ = orc.get_vfm_correspondences(sub, mp, 0.8)
    # indices into the voxelised clouds (RN:288-309)
    src_rows = np.array([np.flatnonzero((scan[:, :3] == sub[i, :3]).all(1))[0] for i in qi])
    corres = np.stack([src_rows, mi], 1).astype(np.int32)
    ref = orc.ransac_corr(scan[:, :3], mp[:, :3], corres, 10000.0, 5000, seed=42)
    np.testing.assert_array_equal(pose, ref.transformation)
    rte, rre = compute_errors(pose, p["T_gt"])
    assert rte < 0.3 and rre < 1.5                                  # RN:973-977 tightest success threshold
    assert compute_errors(pose, p["T_gt"]) == orc.compute_errors(pose, p["T_gt"])
    np.testing.assert_allclose(orthogonalize_rotation(pose), orc.orthogonalize_rotation(pose), atol=0)
    with pytest.raises(ValueError, match="Invalid method"):
        node.ransac_registration(voxel_map, raw_scan, "fpfh")


def test_ransac_registration_retries_with_the_1m_subset(orc):
    """RN:420-423: fewer than 75 correspondences from the 5 m voxel subset -> the 1 m subset is matched instead.  A small
    scan (so that the 5 m subset holds < 75 inlier rows) drives product and oracle re-enactment through the retry; both
    must take it and end with the same pose.  Also runs the ICP refinement of RN:331-344 on the result."""
    from vfmreg import o3d
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import RegistrationNode
    VoxelHashMap.quiet = True
    voxel_map, raw_scan, p = _scene(n_scan=20000, n_map=30000, seed=21)
    # a scan confined to a 24 m ball: a few dozen 5 m voxels, but a thousand 1 m voxels
    raw_scan = raw_scan[np.linalg.norm(raw_scan[:, :3] - raw_scan[0, :3], axis=1) < 12.0]
    assert 300 < len(raw_scan) < 4000
    node = RegistrationNode(ransac_iterations=3000)
    o3d.utility.random.seed(42)
    pose, pose_icp = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True)
    # did the first attempt really fall short?  (oracle pieces, container order)
    scan = orc.voxel_down_sample(orc.voxel_down_sample(raw_scan, 0.5), 1.0)
    mp = voxel_map[orc.voxel_hash_map_points(voxel_map, 1.0, 20)]
    _, _, qi, _, _ = orc.get_vfm_correspondences(orc.voxel_down_sample(scan, 5.0), mp, 0.8)
    assert 3 <= len(qi) < 75
    ref_pose, ref_icp, corres = orc.ransac_registration_vfm(voxel_map, raw_scan, n_iter=3000, run_icp=True)
    assert len(corres) >= 75
    np.testing.assert_array_equal(pose, ref_pose)
    np.testing.assert_array_equal(pose_icp, ref_icp)
    assert np.linalg.norm(pose_icp - p["T_gt"]) < 0.05


def test_find_correspondences_mutual_filter(orc):
    from vfmreg.registration import find_correspondences
    rng = np.random.default_rng(6)
    f1 = rng.standard_normal((700, 33)).astype(np.float32)
    f0 = f1[rng.permutation(700)[:300]] + 0.01 * rng.standard_normal((300, 33)).astype(np.float32)
    for mutual in (True, False):
        i0, i1 = find_correspondences(f0, f1, n_points=100, mutual_filter=mutual)
        r0, r1 = orc.find_correspondences(f0, f1, n_points=100, mutual_filter=mutual)
        np.testing.assert_array_equal(np.sort(i0), np.sort(r0))
        np.testing.assert_array_equal(i1[np.argsort(i0)], r1[np.argsort(r0)])


def test_transform_pcl_mirror(golden):
    from vfmreg.utils import transform_pcl
    g = golden("transform_pcl.npz")
    out = transform_pcl(g["pcl"], g["T"])
    assert out.dtype == np.float32 and out.shape == g["pcl"].shape
    np.testing.assert_allclose(out, g["out32"], rtol=0, atol=1e-5)
    np.testing.assert_array_equal(out[:, 3:], g["pcl"][:, 3:])


# ------------------------------------------------------------------ create_descriptors (rows A2+A3)
@pytest.fixture(params=["one_launch", "per_camera"])
def lift_path(request):
    """create_descriptors has two device paths: projection fused with the gather for all cameras in one launch
    (vfm_lift_multicam) and project -> compact -> gather per camera; both must reproduce the reference fixtures"""
    from vfmreg import prepare_scenes as PS
    PS._FORCE_PER_CAMERA = request.param == "per_camera"
    yield request.param
    PS._FORCE_PER_CAMERA = False


class _UpsampledFeatures:
    """the reference's feature generator seen from create_descriptors: H x W x C after F.interpolate"""

    def __init__(self, grids):
        self.grids, self.calls = grids, 0

    def get_image_features(self, image, upsample=False, cache_file=""):
        g = torch.from_numpy(self.grids[self.calls]).permute(2, 0, 1).unsqueeze(0)
        self.calls += 1
        f = torch.nn.functional.interpolate(g, image.shape[:2], mode="bilinear", align_corners=False)
        return f.squeeze().permute(1, 2, 0).numpy()


class _PatchGridFeatures:
    """fused path: hands the patch grids to the gather kernel (what ImageFeatureGenerator does)"""

    def __init__(self, grids):
        self.grids = grids

    def patch_features_device(self, images):
        return torch.from_numpy(np.stack(self.grids)).cuda()


@pytest.mark.parametrize("fused", [False, True])
def test_create_descriptors_oxford_fixture(golden, fused, lift_path):
    from vfmreg.dataloader import OxfordRobotcar
    from vfmreg.prepare_scenes import create_descriptors
    g = golden("lift_oxf.npz")
    cams = ["stereo/centre", "mono_left", "mono_right"]
    calib = {"lidar_in_ego": g["lidar_in_ego"]}
    cm = {}
    for i, c in enumerate(cams):
        calib[f"{c}_in_ego"] = g["cam_in_ego"][i]
        cm[c] = types.SimpleNamespace(G_camera_image=g["G"], focal_length=(g["fc"][0], g["fc"][1]),
                                      principal_point=(g["fc"][2], g["fc"][3]))
    seq = OxfordRobotcar(calib, cm, image_subsample=int(g["subsample"]), cameras=cams)
    images = {c: g["images"][i] for i, c in enumerate(cams)}
    seq.read_images = lambda filenames=None: images
    fg = (_PatchGridFeatures if fused else _UpsampledFeatures)(list(g["grids"]))
    desc = create_descriptors(None, seq, fg, g["xyz"])
    ref = g["desc"]
    assert desc.dtype == np.float32 and desc.shape == ref.shape
    np.testing.assert_array_equal(np.abs(desc).sum(1) > 0, np.abs(ref).sum(1) > 0)
    np.testing.assert_allclose(desc, ref, rtol=0, atol=1e-5)
    if not fused:  # gathering from the already-upsampled map is a pure copy: bit-exact
        np.testing.assert_array_equal(desc, ref)


@pytest.mark.parametrize("fused", [False, True])
def test_create_descriptors_nclt_fixture(golden, fused, lift_path):
    from vfmreg.dataloader import NCLT
    from vfmreg.prepare_scenes import create_descriptors
    g = golden("lift_nclt.npz")
    cams = ["Cam1", "Cam2"]
    seq = NCLT({c: {"K": g["K"][i], "x_lb3": g["x_lb3"][i]} for i, c in enumerate(cams)},
               {c: {"coords": list(g["coords"])} for c in cams}, image_subsample=int(g["subsample"]), cameras=cams)
    images = {c: g["images"][i] for i, c in enumerate(cams)}
    seq.read_images = lambda filenames=None: images
    fg = (_PatchGridFeatures if fused else _UpsampledFeatures)(list(g["grids"]))
    desc = create_descriptors(None, seq, fg, g["xyz"])
    ref = g["desc"]
    np.testing.assert_array_equal(np.abs(desc).sum(1) > 0, np.abs(ref).sum(1) > 0)
    np.testing.assert_allclose(desc, ref, rtol=0, atol=1e-5)


def test_project_pcl_to_image_mirror_signature(golden):
    from vfmreg.dataloader import KittiOdometry
    g = golden("proj_kitti.npz")
    ds = KittiOdometry({"P2": g["P2"], "Tr_velo_to_cam": g["Tr"]}, image_subsample=1)
    u, v, idx = ds.project_pcl_to_image(g["pcl"], np.zeros((int(g["H"]), int(g["W"]), 3), np.uint8), "camera")
    assert u.dtype == np.int64 and idx.dtype == np.int64
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(u, g["u"])
    np.testing.assert_array_equal(v, g["v"])


def test_image_feature_generator(orc):
    from vfmreg import vit as V
    from vfmreg.image_features import ImageFeatureGenerator
    w = V.random_weights(seed=9, dim=128, depth=2, mlp=256)
    gen = ImageFeatureGenerator("dinov2", use_featup=False, weights=w)
    rng = np.random.default_rng(0)
    image = rng.integers(0, 256, (120, 160, 3), dtype=np.uint8)
    feats = gen.get_image_features(image)                       # 16 x pw x C
    assert feats.shape == (16, gen.patch_w, 128) and gen.patch_w == int((224 / 120) * 160 / 14)
    ref = orc.vit_reference(w, image[None])[0]
    assert np.abs(feats - ref).max() < 1e-2
    up = gen.get_image_features(image, upsample=True)           # H x W x C (IF:104-110)
    assert up.shape == (120, 160, 128)
    t = torch.from_numpy(feats).permute(2, 0, 1).unsqueeze(0)
    up_ref = torch.nn.functional.interpolate(t, (120, 160), mode="bilinear", align_corners=False)
    np.testing.assert_allclose(up, up_ref.squeeze().permute(1, 2, 0).numpy(), rtol=0, atol=1e-5)
    with pytest.raises(ValueError, match="Unsupported foundation model"):
        ImageFeatureGenerator("resnet", use_featup=False)
    with pytest.raises(NotImplementedError):
        ImageFeatureGenerator("dinov2", use_featup=True)


def test_evaluation_harness_on_a_synthetic_scene(tmp_path):
    """row F4 (+ the npz stand-in for the HDF5 scene files, row F3): build the map from posed clouds,
    register every scan with VFM + RANSAC + ICP, recall table as RN:961-989"""
    from vfmreg import synth
    from vfmreg.evaluation import Evaluation, evaluate_scene, read_scenes, save_scene
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import RegistrationNode
    VoxelHashMap.quiet = True
    rng = np.random.default_rng(2)
    d = 128
    world = np.c_[rng.uniform(-40, 40, (40000, 2)), rng.uniform(-2, 6, 40000)]
    desc = np.abs(rng.standard_normal((40000, d))).astype(np.float32)
    desc[::50] = 0.0                                          # points without descriptor are dropped (RN:562)
    map_poses, map_clouds = [], []
    for j in range(3):                                        # three posed map clouds covering the world
        T = synth.random_pose(rng)
        sel = np.arange(j, 40000, 3)
        local = (world[sel] - T[:3, 3]) @ T[:3, :3]           # cloud in its own sensor frame
        map_poses.append(T)
        map_clouds.append(np.c_[local, desc[sel]].astype(np.float32))
    scan_poses, scan_clouds = [], []
    for s in range(2):
        T = synth.random_pose(rng)
        sel = rng.permutation(40000)[:6000]
        sel = sel[desc[sel].sum(1) > 0]
        local = (world[sel] - T[:3, 3]) @ T[:3, :3] + rng.normal(0, 0.01, (len(sel), 3))
        noisy = desc[sel] + 0.05 * np.abs(rng.standard_normal((len(sel), d))).astype(np.float32)
        scan_poses.append(T)
        scan_clouds.append(np.c_[local, noisy].astype(np.float32))
    f = tmp_path / "scene_000.h5"
    save_scene(f, ["mapseq", "scanA", "scanB"], map_poses, map_clouds, scan_poses, scan_clouds)
    scene = read_scenes(f)
    assert len(scene["map_poses"]) == 3 and scene["scene_sequences"] == ["scanA", "scanB"]
    ev = evaluate_scene(scene, RegistrationNode(ransac_iterations=4000), Evaluation())
    assert set(ev.rot_errors) == {"vfm_ransac", "vfm_ransac_icp"} and len(ev.rot_errors["vfm_ransac"]) == 2
    assert ev.compute_success_rate("vfm_ransac_icp", .3, 15) == 1.0
    assert ev.compute_success_rate("vfm_ransac_icp", .6, 1.5) == 1.0
    assert max(ev.trans_errors["vfm_ransac_icp"]) < 0.1
    assert "vfm_ransac_icp" in ev.summary()
    # ---- parity with the oracle's restatement of the same harness (RN:556-593, 858-882, 943-951, 997-1025):
    # the accumulated map (two voxelisation levels in CONTAINER order, fp32 round trips), every pose, every error
    from oracle import oracle as orc
    from vfmreg.evaluation import build_local_map
    ref = orc.evaluate_scene(scene, n_iter=4000)
    np.testing.assert_array_equal(build_local_map(scene["map_poses"], scene["map_point_clouds"], n_descriptors=d), ref["local_map"])
    assert ev.points_in_map == [len(ref["local_map"])] * 2
    for k in ("vfm_ransac", "vfm_ransac_icp"):
        assert ev.rot_errors[k] == ref["rot_errors"][k] and ev.trans_errors[k] == ref["trans_errors"][k], k
    assert ev.error_string().startswith("vfm_ransac\t")


def test_kept_map_searches_take_the_half_width_pass_only_where_it_prunes(orc):
    """VoxelHashMap.search_device on a map that is searched repeatedly: the first search probes the half-width coarse pass
    (vfm_match_search_probe_half); descriptors whose matches stand clear of the background take it from then on, descriptors that are
    all alike (every chunk survives the half-width bound) stay on the full-width pass -- and a map that was probed with the one kind and
    is then searched with the other switches on the search's own load figure.  Every search equals the oracle's."""
    import torch
    from vfmreg import synth
    from vfmreg.mapping import VoxelHashMap
    VoxelHashMap.quiet = True
    rng = np.random.default_rng(3)
    d = 384
    # (a) planted matches on random descriptors
    p = synth.make_pair(1500, 40000, d, seed=5)
    vm = VoxelHashMap(0.01, 1.0e9, 20)
    vm.add_points(np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32))
    mp = vm.point_cloud_n()
    q = p["q_desc"].astype(np.float32)
    qd = torch.from_numpy(q).cuda()
    for _ in range(2):
        qi, mi, _ = vm.search_device(None, 0.8, q_desc=qd)
        _, _, qr, mr, _ = orc.get_vfm_correspondences(np.c_[p["q_xyz"], q], mp, 0.8)
        np.testing.assert_array_equal(qi.cpu().numpy(), qr)
        np.testing.assert_array_equal(mi.cpu().numpy(), mr)
    assert vm._half is True
    # (b) the same map searched by descriptors that are all alike (a large common component): the load figure of the half-width search
    #     is over the limit -- the guard decides that search, the next ones run at full width
    common = rng.standard_normal(d).astype(np.float32)
    alike = (0.25 * rng.standard_normal((1500, d)) + common).astype(np.float32)
    vm2 = VoxelHashMap(0.01, 1.0e9, 20)
    bd = (0.25 * rng.standard_normal((40000, d)) + common).astype(np.float32)
    bx = rng.uniform(-500, 500, (40000, 3)).astype(np.float32)
    vm2.add_points(np.c_[bx, bd])
    mp2 = vm2.point_cloud_n()
    ad = torch.from_numpy(alike).cuda()
    _, _, qr, mr, _ = orc.get_vfm_correspondences(np.c_[np.zeros((1500, 3), np.float32), alike], mp2, 0.8)
    assert len(qr) > 1000
    qi, mi, _ = vm2.search_device(None, 0.8, q_desc=ad)
    assert vm2._half is False                       # probed: no half-width pass on this map
    np.testing.assert_array_equal(qi.cpu().numpy(), qr)
    np.testing.assert_array_equal(mi.cpu().numpy(), mr)
    vm2._half = True                                # (as if the probe had seen a scan that prunes)
    for expect in (False, False):
        qi, mi, _ = vm2.search_device(None, 0.8, q_desc=ad)
        np.testing.assert_array_equal(qi.cpu().numpy(), qr)
        np.testing.assert_array_equal(mi.cpu().numpy(), mr)
        assert vm2._half is expect


def test_ransac_registration_keeps_the_scene_map_between_scans(orc):
    """RegistrationNode keeps the VoxelHashMap it built for a map array while the caller passes the same array (RN:556-589: one
    local_map per scene, many scans): the warm call returns the bits of a cold one; another array, or an edit of the array, rebuilds."""
    from vfmreg import o3d
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import RegistrationNode
    VoxelHashMap.quiet = True
    voxel_map, raw_scan, p = _scene(n_scan=5000, n_map=24000, seed=21)
    _, raw_scan2, _ = _scene(n_scan=5000, n_map=24000, seed=21)
    node, cold = RegistrationNode(ransac_iterations=3000, cache_map=True), RegistrationNode(ransac_iterations=3000)   # off by default
    poses = []
    for nd in (node, node, cold):
        o3d.utility.random.seed(42)
        poses.append(nd.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True))
    built = node._map_cache[2]
    assert cold._map_cache is None
    for a, b in zip(poses[0], poses[1]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(poses[0], poses[2]):
        np.testing.assert_array_equal(a, b)
    ref_pose, ref_icp, _ = orc.ransac_registration_vfm(voxel_map, raw_scan, n_iter=3000, run_icp=True)
    np.testing.assert_array_equal(poses[1][0], ref_pose)
    np.testing.assert_array_equal(poses[1][1], ref_icp)
    node.ransac_registration(voxel_map, raw_scan2, "vfm")
    assert node._map_cache[2] is built                       # another scan of the scene: the same map object
    other = voxel_map.copy()
    node.ransac_registration(other, raw_scan, "vfm")
    assert node._map_cache[2] is not built                   # another array: rebuilt
    built2 = node._map_cache[2]
    other[0, 5] += 1.0                                       # an in-place edit the fingerprint sees (first row)
    node.ransac_registration(other, raw_scan, "vfm")
    assert node._map_cache[2] is not built2
    # ... and one it does NOT see: a single element off the fingerprint's ~4096 strided samples and off the first / last
    # rows -- the stale map is searched.  That is why the cache is opt-in; invalidate_map() is the caller's remedy
    built3 = node._map_cache[2]
    step = max(1, other.size // 4096)
    flat = 1 * other.shape[1] + 1                            # row 1, column 1
    assert flat % step != 0 and flat // other.shape[1] not in (0, other.shape[0] - 1)
    other[1, 1] += 1.0
    node.ransac_registration(other, raw_scan, "vfm")
    assert node._map_cache[2] is built3                      # undetected, by construction of the fingerprint
    node.invalidate_map()
    node.ransac_registration(other, raw_scan, "vfm")
    assert node._map_cache[2] is not built3


def test_set_map_handle_is_the_explicit_form_of_the_kept_map(orc):
    """RegistrationNode.set_map(voxel_map) -> MapHandle: the scene's map built once and passed in place of the array
    (RN:556-589 builds local_map once per scene).  Every method that takes ``voxel_map`` takes the handle; the answers are the cold
    call's and the oracle's bit for bit; the array may be edited or dropped afterwards (the handle owns device copies); a handle is
    independent of cache_map and of other handles."""
    from vfmreg import o3d
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import MapHandle, RegistrationNode
    VoxelHashMap.quiet = True
    voxel_map, raw_scan, p = _scene(n_scan=5000, n_map=24000, seed=33)
    voxel_map_b, raw_scan_b, _ = _scene(n_scan=5000, n_map=24000, seed=34)
    node = RegistrationNode(ransac_iterations=3000)
    o3d.utility.random.seed(42)
    cold = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True)
    h = node.set_map(voxel_map)
    hb = node.set_map(voxel_map_b)
    assert isinstance(h, MapHandle) and (h.rows, h.cols) == voxel_map.shape and node._map_cache is None
    keep = voxel_map.copy()
    voxel_map[:] = 0.0                                      # the caller's array is no longer needed
    for _ in range(2):
        o3d.utility.random.seed(42)
        warm = node.ransac_registration(h, raw_scan, "vfm", run_icp=True)
        for a, b in zip(cold, warm):
            np.testing.assert_array_equal(a, b)
    ref_pose, ref_icp, _ = orc.ransac_registration_vfm(keep, raw_scan, n_iter=3000, run_icp=True)
    np.testing.assert_array_equal(warm[0], ref_pose)
    np.testing.assert_array_equal(warm[1], ref_icp)
    # the other scene's handle, interleaved: each answers for its own map
    o3d.utility.random.seed(42)
    pb = node.ransac_registration(hb, raw_scan_b, "vfm")
    ref_b, _, _ = orc.ransac_registration_vfm(voxel_map_b, raw_scan_b, n_iter=3000, run_icp=False)
    np.testing.assert_array_equal(pb[0], ref_b)
    # compute_vfm_correspondences takes it too
    s1, t1 = node.compute_vfm_correspondences(h, raw_scan)
    s2, t2 = RegistrationNode(ransac_iterations=3000).compute_vfm_correspondences(keep, raw_scan)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(t1, t2)
    with pytest.raises(ValueError):
        node.set_map(np.zeros((5, 2)))


def test_scene_level_descriptor_builder_equals_the_per_cloud_calls(tmp_path):
    """prepare_scenes.main's loop over the clouds of a scene (PS:110-171), with the ViT batched over clouds x cameras: every
    cloud's descriptors are the bits of create_descriptors on that cloud alone, and the scene file read back holds them."""
    from tests.test_gpu_e2e import _cameras
    from tests.test_gpu_vit import _smooth_images
    from vfmreg import vit as V
    from vfmreg.dataloader import KittiOdometry
    from vfmreg.evaluation import read_scenes
    from vfmreg.image_features import ImageFeatureGenerator
    from vfmreg.prepare_scenes import create_descriptors, create_descriptors_batch, prepare_scene

    rng = np.random.default_rng(23)
    H, W, ncam, nclouds = 560, 700, 3, 5
    cams = [f"cam{i}" for i in range(ncam)]
    Ps = _cameras()[:ncam]
    for P in Ps:   # the 1600 x 1200 intrinsics of the C3 rig, scaled to the small test images
        P[0] *= W / 1600.0
        P[1] *= H / 1200.0
    all_imgs = _smooth_images(rng, nclouds * ncam + ncam, H, W)
    clouds = [np.c_[rng.uniform(-40, 40, 3000 + 100 * i), rng.uniform(-40, 40, 3000 + 100 * i), rng.uniform(-2.5, 6, 3000 + 100 * i)]
              .astype(np.float32) for i in range(nclouds + 1)]

    class Seq:
        cameras = cams
        image_subsample = 1

        def __init__(self, name="s", root=None, high_level_api=True):
            self._k = {c: KittiOdometry({"P2": Ps[i], "Tr_velo_to_cam": np.eye(4)}) for i, c in enumerate(cams)}

        def read_pcl(self, filename=None):
            return clouds[int(str(filename).rsplit("_", 1)[1])]

        def read_images(self, filenames=None):
            j = int(str(filenames[0]).rsplit("_", 1)[1])
            return {c: all_imgs[j * ncam + k] for k, c in enumerate(cams)}

        def project_pcl_to_image(self, pcl, image, camera, _device_inputs=None):
            return self._k[camera].project_pcl_to_image(pcl, image, "camera", _device_inputs=_device_inputs)

        def projection_params(self, camera, image_shape):
            return self._k[camera].projection_params("camera", image_shape)

    gen = ImageFeatureGenerator("dinov2", use_featup=False, weights=V.random_weights(seed=3, dim=384, depth=2, mlp=1536))
    seq = Seq()
    files = [[f"img_{i}"] for i in range(nclouds)]
    single = [create_descriptors(files[i], seq, gen, clouds[i]) for i in range(nclouds)]
    for per_forward in (2, 8):
        batched = create_descriptors_batch(files, seq, gen, clouds[:nclouds], clouds_per_forward=per_forward)
        for a, b in zip(single, batched):
            np.testing.assert_array_equal(a, b)
    assert all((np.abs(d).sum(1) > 0).sum() > 500 for d in single)
    scene = {"mapping": {"point_clouds": [f"x/mapseq/cloud_{i}" for i in range(nclouds)], "images": [[f"img_{i}"] for i in range(nclouds)],
                         "poses": [np.eye(4).tolist()] * nclouds},
             "registration": [{"point_cloud": f"x/scanseq/cloud_{nclouds}", "images": [f"img_{nclouds}"], "pose": np.eye(4).tolist()}]}
    out = tmp_path / "scene.h5"
    sequences, map_poses, map_clouds, seq_poses, seq_clouds = prepare_scene(tmp_path, scene, Seq, gen, date_idx=1, output_filename=out,
                                                                             voxel_down_sample=lambda p, v: p)
    assert sequences == ["mapseq", "scanseq"] and len(map_clouds) == nclouds and len(seq_clouds) == 1
    for i in range(nclouds):
        np.testing.assert_array_equal(map_clouds[i][:, 3:], single[i])
    back = read_scenes(out)
    for a, b in zip(back["map_point_clouds"], map_clouds):
        np.testing.assert_array_equal(a, b)
