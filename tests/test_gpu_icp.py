"""Row F2 on the GPU: ICP nearest-neighbour + normal-equation kernels and the register_frame mirror
(kiss_icp RegisterFrame, Registration.cpp:145-195) against the CPU oracle.  The kernels replay the
oracle's operation order and reduction tree, so every iterate -- and the final pose -- is bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(seed=5, n=3000, m=30000):
    from oracle import oracle as orc
    from vfmreg import synth
    p = synth.make_pair(n, m, 128, seed=seed)
    mp = p["b_xyz"][orc.voxel_first(p["b_xyz"], 1.0, 20)]
    return p, mp


def test_icp_kernels_bit_exact():
    import ctypes as C
    from oracle import oracle as orc
    from vfmreg import _lib, ops
    from vfmreg.icp import VoxelGridDevice
    lib = _lib.load()
    p, mp = _scene()
    T0 = p["T_gt"].copy()
    T0[:3, 3] += [0.3, -0.2, 0.1]
    src = orc.transform_pcl(p["q_xyz"], T0)
    src[5] = [500.0, 500.0, 500.0]        # far from every voxel: no neighbour -> invalid
    src[6] = mp[17] + [0.0, 0.0, 1e-9]    # practically on a map point
    keys, start, pts = orc.voxel_grid_csr(mp, 1.0)
    g = VoxelGridDevice(mp, 1.0)
    np.testing.assert_array_equal(g.keys.cpu().numpy(), keys)
    np.testing.assert_array_equal(g.start.cpu().numpy(), start)
    n = len(src)
    for max_dist in (6.0, 0.4):
        tgt_r = np.empty_like(src)
        val_r = np.empty(n, dtype=np.uint8)
        orc.lib().orc_icp_nearest(orc._p(src, orc._f64p), C.c_int64(n), orc._p(keys, orc._i64p), orc._p(start, orc._i32p),
                                  orc._p(pts, orc._f64p), C.c_int32(len(keys)), C.c_double(1.0), C.c_double(max_dist),
                                  orc._p(tgt_r, orc._f64p), orc._p(val_r, orc._u8p))
        s_d = dev(src)
        tgt = torch.empty_like(s_d)
        val = torch.empty(n, dtype=torch.uint8, device="cuda")
        _lib.check(lib.vfm_icp_nearest(s_d.data_ptr(), n, g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(),
                                       g.n_voxels, 1.0, max_dist, tgt.data_ptr(), val.data_ptr(), ops._stream()))
        np.testing.assert_array_equal(val.cpu().numpy(), val_r)
        np.testing.assert_array_equal(tgt.cpu().numpy()[val_r > 0], tgt_r[val_r > 0])
        assert val_r[5] == 0 and val_r[6] == 1 and 0 < val_r.sum() <= n
        # brute-force check of the neighbour search itself (all map points within reach are in the 27 voxels
        # only if they are closer than one voxel: compare where the true NN is within 1 m)
        d = np.linalg.norm(mp[None, :, :] - src[:200, None, :], axis=2)
        nn = d.argmin(1)
        close = d.min(1) < min(max_dist, 1.0)
        np.testing.assert_array_equal(tgt_r[:200][close], mp[nn[close]])
        out_r = np.empty(43)
        orc.lib().orc_icp_system(orc._p(src, orc._f64p), orc._p(tgt_r, orc._f64p), orc._p(val_r, orc._u8p),
                                 C.c_int64(n), C.c_double(2 / 3), orc._p(out_r, orc._f64p))
        out = torch.empty(43, dtype=torch.float64, device="cuda")
        _lib.check(lib.vfm_icp_build_system(s_d.data_ptr(), tgt.data_ptr(), val.data_ptr(), n, 2 / 3, out.data_ptr(),
                                            ops._stream()))
        np.testing.assert_array_equal(out.cpu().numpy(), out_r)
        assert out_r[42] == val_r.sum()
        np.testing.assert_allclose(out_r[:36].reshape(6, 6), out_r[:36].reshape(6, 6).T, rtol=1e-12)


def test_register_frame_matches_oracle_and_refines_pose():
    from oracle import oracle as orc
    from vfmreg.config import load_config
    from vfmreg.icp import register_frame
    from vfmreg.mapping import get_voxel_hash_map
    p, mp = _scene(seed=8)
    cfg = load_config(None, None)
    voxel_hash_map = get_voxel_hash_map(cfg)
    voxel_hash_map.add_points(p["b_xyz"])
    rng = np.random.default_rng(1)
    guess = p["T_gt"].copy()
    guess[:3, 3] += rng.normal(0, 0.25, 3)
    sigma = cfg.adaptive_threshold.initial_threshold
    pose = register_frame(points=p["q_xyz"], voxel_map=voxel_hash_map, initial_guess=guess,
                          max_correspondance_distance=3 * sigma, kernel=sigma / 3)      # RN:340-344
    ref = orc.register_frame(p["q_xyz"], voxel_hash_map.point_cloud(), cfg.mapping.voxel_size, guess, 3 * sigma, sigma / 3)
    assert np.linalg.norm(pose - ref) <= 1e-5
    np.testing.assert_array_equal(pose, ref)
    assert np.linalg.norm(pose - p["T_gt"]) < 0.02 < np.linalg.norm(guess - p["T_gt"])
    # empty map -> the initial guess comes back (Registration.cpp:150); wrong widths are loud
    empty = get_voxel_hash_map(cfg)
    np.testing.assert_array_equal(register_frame(p["q_xyz"], empty, guess, 6.0, 0.6), guess)
    with pytest.raises(ValueError, match="Invalid shape"):
        register_frame(np.zeros((4, 2)), voxel_hash_map, guess, 6.0, 0.6)


def test_ransac_registration_with_icp_refinement():
    from vfmreg import o3d, synth
    from vfmreg.mapping import VoxelHashMap
    from vfmreg.registration import RegistrationNode, compute_errors
    VoxelHashMap.quiet = True
    p = synth.make_pair(6000, 30000, 384, seed=11)
    voxel_map = np.c_[p["b_xyz"], p["b_desc"]]
    raw_scan = np.c_[p["q_xyz"], p["q_desc"]]
    node = RegistrationNode(ransac_iterations=5000)
    o3d.utility.random.seed(42)
    ransac_pose, pose = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True)   # RN:867-875
    e_ransac, e_icp = compute_errors(ransac_pose, p["T_gt"]), compute_errors(pose, p["T_gt"])
    assert abs(np.linalg.det(ransac_pose[:3, :3]) - 1) <= 1e-12          # orthogonalised (RN:331-336)
    assert e_icp[0] < 0.05 and e_icp[1] < 0.05 and e_icp[0] <= e_ransac[0] + 1e-3


def test_icp_nearest_keeps_the_first_minimum_of_the_reference_scan_on_exact_ties():
    """vfm_icp_nearest spreads a point's 27 neighbour voxels over 32 lanes and merges their candidates by (squared distance, scan
    position).  On a lattice map a source point at a cell centre / face centre / lattice point is EXACTLY equidistant from points
    in several voxels, and doubled map points tie inside one voxel: the answer must be the first minimum of the reference's scan
    (voxel loops i, j, k ascending, then insertion order; VoxelHashMap.cpp:96-130) -- the oracle's, bit for bit, negative
    coordinates (truncation towards zero in the voxel index) included."""
    import ctypes as C
    from oracle import oracle as orc
    from vfmreg import _lib, ops
    from vfmreg.icp import VoxelGridDevice
    lib = _lib.load()
    rng = np.random.default_rng(3)
    ax = np.arange(-6, 7) * 0.5                       # lattice of pitch 0.5 in voxels of 1.0: up to eight points per voxel
    mp = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    mp = np.concatenate([mp, mp[rng.integers(0, len(mp), 300)]])   # doubled points: ties inside a voxel, different insertion positions
    mp = mp[rng.permutation(len(mp))].copy()
    cells = rng.integers(-5, 5, (1500, 3)) * 0.5
    src = np.concatenate([cells + 0.25,                            # cell centres: eight equidistant lattice points
                          cells + np.array([0.25, 0.25, 0.0]),     # face centres: four
                          cells + np.array([0.25, 0.0, 0.0]),      # edge midpoints: two
                          cells])                                  # on a lattice point (possibly a doubled one)
    keys, start, pts = orc.voxel_grid_csr(mp, 1.0)
    g = VoxelGridDevice(mp, 1.0)
    n = len(src)
    tgt_r = np.empty_like(src)
    val_r = np.empty(n, dtype=np.uint8)
    orc.lib().orc_icp_nearest(orc._p(src, orc._f64p), C.c_int64(n), orc._p(keys, orc._i64p), orc._p(start, orc._i32p),
                              orc._p(pts, orc._f64p), C.c_int32(len(keys)), C.c_double(1.0), C.c_double(2.0),
                              orc._p(tgt_r, orc._f64p), orc._p(val_r, orc._u8p))
    s_d = dev(src)
    tgt = torch.empty_like(s_d)
    val = torch.empty(n, dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_icp_nearest(s_d.data_ptr(), n, g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(),
                                   g.n_voxels, 1.0, 2.0, tgt.data_ptr(), val.data_ptr(), ops._stream()))
    np.testing.assert_array_equal(val.cpu().numpy(), val_r)
    assert val_r.all()
    np.testing.assert_array_equal(tgt.cpu().numpy(), tgt_r)
    # the ties are real: at the cell centres several distinct map points are at the minimum distance
    d = np.linalg.norm(mp[None] - src[:50, None], axis=2)
    assert ((d == d.min(1, keepdims=True)).sum(1) >= 8).all()


def test_register_frame_387_columns():
    """The descriptor-seeded RegisterFrame (Registration.cpp:197-382; register_frame on rows of 3 + 384 columns): 5 m subset ->
    GetVFMCorrespondences(0.8) -> Gauss-Newton on the descriptor pairs with median + 1.5 MAD pruning -> vanilla ICP, against the
    oracle's restatement: pose, the surviving pairs (src_ moved by every later update) bit for bit; the pose improves on the guess;
    the bare call returns the pose only (registration.py:47-66); an empty descriptor map hands the guess back."""
    from oracle import oracle as orc
    from vfmreg import synth
    from vfmreg.config import load_config
    from vfmreg.icp import register_frame
    from vfmreg.mapping import VoxelHashMap, get_voxel_hash_map
    VoxelHashMap.quiet = True
    cfg = load_config(None, None)
    sigma = cfg.adaptive_threshold.initial_threshold
    for seed, n_scan, n_map, shift in ((9, 6000, 30000, 0.4), (10, 900, 12000, 0.15)):   # the second: fewer than 100 voxels of 5 m -> whole scan
        p = synth.make_pair(n_scan, n_map, 384, seed=seed)
        voxel_map = np.c_[p["b_xyz"], p["b_desc"]]
        scan = np.c_[p["q_xyz"], p["q_desc"]]
        if seed == 10:
            keep = np.linalg.norm(scan[:, :3] - scan[:, :3].mean(0), axis=1) < 18.0       # a compact scan: few 5 m voxels
            scan = scan[keep]
        vhm = get_voxel_hash_map(cfg)
        vhm.add_points(voxel_map)
        rng = np.random.default_rng(seed)
        guess = p["T_gt"].copy()
        guess[:3, 3] += rng.normal(0, shift, 3)
        pose, src_, tgt_ = register_frame(scan, vhm, guess, 3 * sigma, sigma / 3, src_=np.zeros((1, 3)), tgt_=np.zeros((1, 3)))
        ref, rs, rt, hist = orc.register_frame_nd(scan, vhm.point_cloud_n(), cfg.mapping.voxel_size, guess, 3 * sigma, sigma / 3,
                                                  return_history=True)
        assert any(h[0] == "vfm" for h in hist) and any(h[0] == "icp" for h in hist)
        np.testing.assert_array_equal(pose, ref)
        np.testing.assert_array_equal(src_, rs)
        np.testing.assert_array_equal(tgt_, rt)
        assert len(src_) > 10 and np.linalg.norm(pose - p["T_gt"]) < 0.03 < np.linalg.norm(guess - p["T_gt"])
        only = register_frame(scan, vhm, guess, 3 * sigma, sigma / 3)
        np.testing.assert_array_equal(only, ref)
    empty = get_voxel_hash_map(cfg)
    np.testing.assert_array_equal(register_frame(scan, empty, guess, 6.0, 0.6), guess)
    with pytest.raises(NotImplementedError):
        register_frame(scan[:, :200], vhm, guess, 6.0, 0.6)
