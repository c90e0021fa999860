"""The CPU oracle is pinned against outputs of the reference's own functions (tests/golden/*.npz,
produced by tests/golden/make_golden.py from /root/reference in the build container)."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_projection_nclt_matches_reference(golden):
    g = golden("proj_nclt.npz")
    sub = float(g["subsample"])
    win = g["coords"] // int(sub)
    img = g["image"]
    u, v, idx = orc.project(0, g["pcl"], [g["T_c_body"], g["K"]], None, sub, win, img,
                            img.shape[0], img.shape[1])
    assert len(idx) == len(g["idx"]) > 1000
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(u, g["u"])
    np.testing.assert_array_equal(v, g["v"])


def test_projection_oxford_matches_reference(golden):
    g = golden("proj_oxf.npz")
    u, v, idx = orc.project(1, g["pcl"], [g["lidar_in_ego"], g["cam_in_ego"], g["Ginv"]], g["fc"],
                            float(g["subsample"]), None, None, int(g["H"]), int(g["W"]))
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(u, g["u"])
    np.testing.assert_array_equal(v, g["v"])


def test_projection_kitti_matches_reference(golden):
    g = golden("proj_kitti.npz")
    u, v, idx = orc.project(2, g["pcl"], [g["P2Tr"]], None, float(g["subsample"]), None, None,
                            int(g["H"]), int(g["W"]))
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(u, g["u"])
    np.testing.assert_array_equal(v, g["v"])


def _lift_oxf_cams(g):
    n = g["xyz"].shape[0]
    pcl = np.insert(g["xyz"], 3, values=1, axis=1).T
    cams = []
    for c in range(g["images"].shape[0]):
        img = g["images"][c]
        H, W = img.shape[:2]
        u, v, idx = orc.project(1, pcl, [g["lidar_in_ego"], g["cam_in_ego"][c], g["Ginv"]], g["fc"],
                                float(g["subsample"]), None, None, H, W)
        cams.append(dict(grid=g["grids"][c], Hup=H, Wup=W, rot_mode=0,
                         black=np.all(img == 0, axis=-1), u=u, v=v, idx=idx))
    return n, cams


def test_create_descriptors_oxford_matches_reference(golden):
    g = golden("lift_oxf.npz")
    n, cams = _lift_oxf_cams(g)
    desc = orc.create_descriptors(n, cams)
    ref = g["desc"]
    # which points carry a descriptor (indices, first-camera-wins, black-pixel zeroing): exact
    np.testing.assert_array_equal(np.abs(desc).sum(1) > 0, np.abs(ref).sum(1) > 0)
    # values: fused "interpolate at the pixel" vs torch's upsample-then-index: fp32 evaluation
    # order differs -> tolerance, stated here: 1e-5 absolute
    np.testing.assert_allclose(desc, ref, rtol=0, atol=1e-5)


def test_create_descriptors_nclt_matches_reference(golden):
    g = golden("lift_nclt.npz")
    n = g["xyz"].shape[0]
    pcl = np.insert(g["xyz"], 3, values=1, axis=1).T
    sub = float(g["subsample"])
    win = g["coords"] // int(sub)
    cams = []
    for c in range(g["images"].shape[0]):
        raw = g["images"][c]                      # H x W x 3 as read
        rot = np.ascontiguousarray(np.rot90(raw, 1))  # PS:73-74 cv2.rotate(..., 90 CCW)
        u, v, idx = orc.project(0, pcl, [g["T_c_body"][c], g["K"][c]], None, sub, win, rot,
                                rot.shape[0], rot.shape[1])
        cams.append(dict(grid=g["grids"][c], Hup=raw.shape[0], Wup=raw.shape[1], rot_mode=1,
                         black=np.all(raw == 0, axis=-1), u=u, v=v, idx=idx))
    desc = orc.create_descriptors(n, cams)
    ref = g["desc"]
    np.testing.assert_array_equal(np.abs(desc).sum(1) > 0, np.abs(ref).sum(1) > 0)
    np.testing.assert_allclose(desc, ref, rtol=0, atol=1e-5)


def test_transform_pcl_matches_reference(golden):
    g = golden("transform_pcl.npz")
    out32 = orc.transform_pcl(g["pcl"], g["T"])
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, g["out32"], rtol=0, atol=1e-5)
    out64 = orc.transform_pcl(g["pcl"].astype(np.float64), g["T"])
    np.testing.assert_allclose(out64, g["out64"], rtol=0, atol=1e-12)


def test_kabsch_matches_reference_rigid_transform_3d(golden):
    g = golden("kabsch_dsc.npz")
    A, B, w = g["A"], g["B"], g["w"]
    for b in range(A.shape[0]):
        T, ok = orc.kabsch(A[b], B[b], None, denom_eps=1e-6)
        assert ok
        # reference runs torch fp32 (pointdsc/common.py:7-47): tolerance 2e-4 on the 4x4
        np.testing.assert_allclose(T, g["T_unw"][b], rtol=0, atol=2e-4)
        Tw, ok = orc.kabsch(A[b], B[b], w[b], denom_eps=1e-6)
        assert ok
        np.testing.assert_allclose(Tw, g["T_w"][b], rtol=0, atol=2e-4)
        T3, ok = orc.kabsch(g["A3"][b], g["B3"][b], None, denom_eps=1e-6)
        assert ok
        np.testing.assert_allclose(T3, g["T_3"][b], rtol=0, atol=5e-3)  # 3-point, fp32 SVD of rank-2 H


def test_kabsch_fixed_sequence_equals_numpy_svd():
    rng = np.random.default_rng(3)
    for n in (3, 4, 10, 200):
        for _ in range(50):
            A = rng.uniform(-20, 20, (n, 3))
            R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
            if np.linalg.det(R) < 0:
                R[:, 0] *= -1
            B = A @ R.T + rng.normal(0, 3, 3) + rng.normal(0, 0.05, (n, 3))
            T, ok = orc.kabsch(A, B)
            assert ok
            np.testing.assert_allclose(T, orc.kabsch_svd(A, B), rtol=0, atol=1e-9)
            assert abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    # degenerate samples are reported invalid, not silently solved
    A = np.array([[0.0, 0, 0], [1, 1, 1], [2, 2, 2]])
    assert not orc.kabsch(A, A + 1.0)[1]
    assert not orc.kabsch(A[[0, 0, 0]], A[[1, 1, 1]])[1]


def test_kabsch_reflection_case_gives_proper_rotation():
    rng = np.random.default_rng(4)
    A = rng.uniform(-5, 5, (30, 3))
    B = A * np.array([1.0, 1.0, -1.0])  # mirror: best proper rotation, det must stay +1
    T, ok = orc.kabsch(A, B)
    assert ok and abs(np.linalg.det(T[:3, :3]) - 1) < 1e-12
    np.testing.assert_allclose(T, orc.kabsch_svd(A, B), rtol=0, atol=1e-8)
