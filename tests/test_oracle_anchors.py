"""Anchors for the oracle's restatements of third-party arithmetic that is absent from the reference tree
(faiss IndexFlatIP, scipy cKDTree as called by find_correspondences, Open3D's correspondence RANSAC with
Eigen::umeyama, kiss-icp's first-point-per-voxel maps): each is compared with an INDEPENDENT implementation
of the published algorithm that exists in this image (numpy fp32 sgemm, scipy.spatial.cKDTree, numpy SVD,
a sequential dict).  Where the oracle fixes something the original leaves open (summation order, ties),
the comparison allows exactly that freedom and nothing else."""
import numpy as np
import pytest

from oracle import oracle as orc


def test_match_oracle_vs_faiss_algorithm_in_fp32():
    """IndexFlatIP.search(k=1) = fvec_renorm_L2 + sgemm + row arg-max (VHM:469-495).  The oracle decides in
    fp64; it may differ from the fp32 product only where the two best fp32 scores are within fp32 rounding."""
    from vfmreg import synth
    p = synth.make_pair(1500, 9000, 384, seed=3)
    q = p["q_desc"] * np.float32(2.5)
    b = p["b_desc"] * np.float32(0.3)
    qn, inv = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    ref_norm = q / np.sqrt((q.astype(np.float64) ** 2).sum(1, keepdims=True))
    assert np.abs(qn - ref_norm).max() < 2e-7                       # fvec_renorm_L2: x / sqrt(sum x^2) in fp32
    s32 = qn @ bn.T                                                  # the sgemm faiss runs
    idx32 = s32.argmax(1)
    idx, sim = orc.match_ip_top1_bruteforce(qn, bn)
    top2 = np.partition(s32, -2, axis=1)[:, -2:]
    near_tie = (top2[:, 1] - top2[:, 0]) < 4e-6
    assert (idx == idx32)[~near_tie].all()
    assert (idx != idx32).sum() <= near_tie.sum()
    assert np.abs(sim - s32.max(1)).max() < 2e-6
    idx_b, sim_b = orc.match_ip_top1(qn, bn)                         # accelerated oracle == brute-force oracle
    np.testing.assert_array_equal(idx_b, idx)
    np.testing.assert_array_equal(sim_b, sim)
    keep = orc.threshold_compact(sim, 0.8)                           # valid = D >= 0.8, query order (VHM:501-511)
    np.testing.assert_array_equal(keep, np.nonzero(~(sim.astype(np.float64) < 0.8))[0])


def test_nn_l2_oracle_vs_ckdtree():
    """find_correspondences' neighbours come from scipy.spatial.cKDTree(...).query(k=1) (RN:486-496)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(4)
    for d in (3, 33, 64):
        a = rng.standard_normal((700, d)).astype(np.float32)
        b = rng.standard_normal((1900, d)).astype(np.float32)
        dist_ref, idx_ref = cKDTree(b.astype(np.float64)).query(a.astype(np.float64), k=1)
        idx, dist = orc.nn_l2(a, b)
        np.testing.assert_array_equal(idx, idx_ref)
        np.testing.assert_allclose(dist, dist_ref, rtol=1e-13, atol=0)


def test_find_correspondences_vs_reference_logic():
    """RN:482-538 restated with the library the reference uses (cKDTree + argpartition / mutual filter)."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    f0 = rng.standard_normal((900, 32)).astype(np.float32)
    f1 = np.r_[f0[rng.permutation(900)[:600]] + 0.05 * rng.standard_normal((600, 32)).astype(np.float32),
               rng.standard_normal((500, 32)).astype(np.float32)]
    d01, nn01 = cKDTree(f1).query(f0, k=1)
    _, nn10 = cKDTree(f0).query(f1, k=1)
    i0 = np.arange(len(f0))
    mutual = nn10[nn01] == i0
    a, b = orc.find_correspondences(f0, f1, mutual_filter=True)
    np.testing.assert_array_equal(a, i0[mutual])
    np.testing.assert_array_equal(b, nn01[mutual])
    n = min(300, len(d01) - 1)
    a, b = orc.find_correspondences(f0, f1, n_points=300, mutual_filter=False)
    assert set(a.tolist()) == set(np.argpartition(d01, n)[:n].tolist())
    np.testing.assert_array_equal(b, nn01[a])


def _umeyama_numpy(A, B):
    """Eigen::umeyama without scaling: R = U diag(1, 1, det) V^T of the cross-covariance, t = mean_B - R mean_A."""
    ma, mb = A.mean(0), B.mean(0)
    H = (B - mb).T @ (A - ma) / len(A)
    U, S, Vt = np.linalg.svd(H)
    D = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        D[2, 2] = -1
    R = U @ D @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mb - R @ ma
    return T


def test_ransac_oracle_vs_numpy_open3d_rule():
    """Open3D's registration_ransac_based_on_correspondence: sample 3 correspondences, umeyama, score all
    correspondences (inlier iff |T s - t| < d; fitness = inliers / C; rmse over inliers), keep the higher
    fitness, then the lower rmse.  Every hypothesis of the oracle is re-derived with numpy's SVD."""
    rng = np.random.default_rng(6)
    from vfmreg import synth
    C, n_iter, seed, dmax = 400, 300, 9, 0.3
    T_gt = synth.random_pose(rng)
    src = np.c_[rng.uniform(-30, 30, C), rng.uniform(-30, 30, C), rng.uniform(-2, 8, C)]
    tgt = src @ T_gt[:3, :3].T + T_gt[:3, 3] + rng.normal(0, 0.03, src.shape)
    bad = rng.random(C) < 0.4
    tgt[bad] = rng.uniform(-30, 30, (int(bad.sum()), 3))
    corres = np.stack([np.arange(C), rng.permutation(C)], 1).astype(np.int32)
    tgt_cloud = np.empty_like(tgt)
    tgt_cloud[corres[:, 1]] = tgt
    r = orc.ransac_corr(src, tgt_cloud, corres, dmax, n_iter, seed=seed, per_hyp=True)
    S, Q = src[corres[:, 0]], tgt_cloud[corres[:, 1]]
    best = None
    checked = 0
    for h in range(n_iter):
        w = orc.philox(h, seed)
        pick = [(int(w[k]) * C) >> 32 for k in range(3)]            # uniform with replacement
        if len(set(pick)) < 3:
            continue                                                  # rank-deficient sample: cannot win
        T = _umeyama_numpy(S[pick], Q[pick])
        d2 = (((S @ T[:3, :3].T + T[:3, 3]) - Q) ** 2).sum(1)
        inl = d2 < dmax * dmax
        fit = inl.sum() / C
        rmse = np.sqrt(d2[inl].sum() / max(inl.sum(), 1))
        if np.abs(d2 - dmax * dmax).min() > 1e-9:                    # no borderline point: counts must agree exactly
            assert r.hyp_fit[h] == fit, h
            assert abs(r.hyp_rmse[h] - rmse) <= 1e-9 * max(rmse, 1e-12), h
            checked += 1
        if fit > 0 and (best is None or fit > best[0] or (fit == best[0] and rmse < best[1])):
            best = (fit, rmse, h, T)
    assert checked > n_iter // 2
    assert r.best_hyp == best[2] and r.fitness == best[0]
    assert np.linalg.norm(r.transformation - best[3]) < 1e-9
    assert np.linalg.norm(r.transformation - T_gt) < 0.1
    d2 = (((S @ r.transformation[:3, :3].T + r.transformation[:3, 3]) - Q) ** 2).sum(1)
    np.testing.assert_array_equal(r.inlier_mask.astype(bool), d2 < dmax * dmax)


def test_voxel_first_oracle_vs_sequential_map():
    """kiss-icp keeps the first point that falls into a voxel (Preprocessing.cpp:50-69) resp. the first
    max_points_per_voxel points (VoxelHashMap.cpp:746-757); voxel = trunc-toward-zero of p / voxel_size."""
    rng = np.random.default_rng(7)
    pts = rng.uniform(-20, 20, (5000, 3))
    pts[100] = pts[7]
    for vs, cap in ((1.0, 1), (0.5, 1), (2.0, 20), (5.0, 3)):
        seen = {}
        keep = []
        for i, p in enumerate(pts):
            key = tuple(np.trunc(p / vs).astype(np.int64))           # static_cast<int>: truncation (PRE.cpp:58)
            if seen.get(key, 0) < cap:
                seen[key] = seen.get(key, 0) + 1
                keep.append(i)
        np.testing.assert_array_equal(orc.voxel_first(pts, vs, cap), np.array(keep))


def test_icp_oracle_recovers_planted_pose():
    """RegisterFrame (Registration.cpp:96-195): point-to-point Gauss-Newton with the 27-voxel neighbourhood
    search must pull a perturbed scan back onto the map."""
    rng = np.random.default_rng(8)
    m = np.c_[rng.uniform(-20, 20, 6000), rng.uniform(-20, 20, 6000), rng.uniform(-1, 3, 6000)]
    from vfmreg import synth
    T = np.eye(4)
    ang = np.deg2rad(1.5)
    T[:3, :3] = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    T[:3, 3] = [0.25, -0.15, 0.05]
    scan = (m[:2500] - T[:3, 3]) @ T[:3, :3]                          # scan = T^-1 map subset
    est = orc.register_frame(scan, m, 1.0, np.eye(4), 3.0, 1.0)
    est = est[0] if isinstance(est, tuple) else est
    assert np.linalg.norm(est - T) < 1e-6


def test_descriptor_seeded_icp_restatement_recovers_the_planted_pose_and_its_median_is_numpys():
    """oracle.register_frame_nd (Registration.cpp:197-382): both stages run, the pose ends within millimetres of the planted one from a
    0.5 m error, the surviving descriptor pairs are a subset of the initial ones moved onto their targets; the reference's
    nth_element median (the mean of the two middle order statistics for even sizes) is numpy's median."""
    from oracle import oracle as orc
    from vfmreg import synth
    rng = np.random.default_rng(2)
    for n in (1, 2, 7, 8, 101, 256):
        v = rng.standard_normal(n)
        assert orc.median_like_the_reference(v) == float(np.median(v))
    p = synth.make_pair(3000, 15000, 384, seed=9)
    vm = np.c_[p["b_xyz"], p["b_desc"]]
    mp = vm[orc.voxel_hash_map_points(vm, 1.0, 20)]
    scan = np.c_[p["q_xyz"], p["q_desc"]]
    guess = p["T_gt"].copy()
    guess[:3, 3] += 0.3
    T, s, t, hist = orc.register_frame_nd(scan, mp, 1.0, guess, 6.0, 2 / 3, return_history=True)
    kinds = [h[0] for h in hist]
    assert "vfm" in kinds and "icp" in kinds and kinds == sorted(kinds, key=lambda k: k != "vfm")   # the VFM stage first
    assert np.linalg.norm(T - p["T_gt"]) < 5e-3 < np.linalg.norm(guess - p["T_gt"])
    assert len(s) == len(t) > 100 and np.linalg.norm(s - t, axis=1).max() < 0.2
