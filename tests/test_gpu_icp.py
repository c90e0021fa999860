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
    with pytest.raises(ValueError):      # rows of a width the map does not hold
        register_frame(scan[:, :200], vhm, guess, 6.0, 0.6)


@pytest.mark.parametrize("width,zero_rows", [(3 + 64, True), (3 + 768, False), (3 + 1, False)])
def test_register_frame_other_widths_take_the_vectorxd_loop(width, zero_rows):
    """RegisterFrame(std::vector<Eigen::VectorXd> ...) (Registration.cpp:384-423; VERDICT r5 "missing" item 4): rows whose width is neither
    3 nor _point_size() = 387 run the 3-D loop with VoxelHashMap::GetCorrespondences(VectorXdVector) (VoxelHashMap.cpp:321-448) as
    its search -- squared distance x clamp(0.5 (1 - cos), 0.01, 1), weight 1 where a descriptor sums to zero, Euclidean acceptance.
    Against the oracle's restatement: the pose bit for bit, every iterate's normal equations and chosen neighbours bit for bit (through
    the kernels: vfm_icp_step_nearest_desc / vfm_icp_desc_stats against orc_icp_nearest_desc / orc_icp_desc_stats); the descriptors
    matter -- with descriptors that contradict the geometry the search picks other neighbours than the plain 3-D one; the pose improves."""
    from oracle import oracle as orc
    from vfmreg import _lib, synth
    from vfmreg.config import load_config
    from vfmreg.icp import _DescGrid, register_frame
    from vfmreg.mapping import VoxelHashMap, get_voxel_hash_map
    VoxelHashMap.quiet = True
    cfg = load_config(None, None)
    sigma = cfg.adaptive_threshold.initial_threshold
    f = width - 3
    rng = np.random.default_rng(width)
    n_map, n_scan = 20000, 3000
    m_xyz = np.c_[rng.uniform(-25, 25, n_map), rng.uniform(-25, 25, n_map), rng.uniform(-2, 4, n_map)]
    m_desc = rng.standard_normal((n_map, f)).astype(np.float32)
    T_gt = synth.random_pose(rng)
    T_gt[:3, 3] *= 0.1
    pick = rng.choice(n_map, n_scan, replace=False)
    R, t = T_gt[:3, :3], T_gt[:3, 3]
    s_xyz = (m_xyz[pick] - t) @ R + rng.normal(0, 0.02, (n_scan, 3))
    s_desc = m_desc[pick] + 0.3 * rng.standard_normal((n_scan, f)).astype(np.float32)
    if zero_rows:
        s_desc[::7] = 0.0                # no descriptor: weight 1 (VHM:366)
        m_desc[::5] = 0.0
        s_desc[3, :2] = [1.0, -1.0]      # a non-zero row whose elements sum to zero: treated as "no descriptor" too
        s_desc[3, 2:] = 0.0
    voxel_map = np.c_[m_xyz, m_desc].astype(np.float64)
    scan = np.c_[s_xyz, s_desc].astype(np.float64)
    vhm = get_voxel_hash_map(cfg)
    vhm.add_points(voxel_map)
    guess = T_gt.copy()
    guess[:3, 3] += rng.normal(0, 0.25, 3)
    pose = register_frame(scan, vhm, guess, 3 * sigma, sigma / 3)
    ref, hist = orc.register_frame_xd(scan, vhm.point_cloud_n(), cfg.mapping.voxel_size, guess, 3 * sigma, sigma / 3, return_history=True)
    np.testing.assert_array_equal(pose, ref)
    assert len(hist) >= 2 and np.linalg.norm(pose - T_gt) < 0.05 < np.linalg.norm(guess - T_gt)
    # one search through the C ABI against the oracle's, the chosen neighbour of every point
    lib = _lib.load()
    g = _DescGrid(vhm.point_cloud_n(), cfg.mapping.voxel_size)
    src = torch.from_numpy(np.ascontiguousarray(scan[:, :3])).cuda()
    sd = torch.from_numpy(np.ascontiguousarray(scan[:, 3:])).cuda()
    sn = torch.empty(n_scan, dtype=torch.float64, device="cuda")
    sh = torch.empty(n_scan, dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_icp_desc_stats(sd.data_ptr(), n_scan, f, sn.data_ptr(), sh.data_ptr(), None))
    moved = torch.empty_like(src)
    tgt = torch.empty_like(src)
    valid = torch.empty(n_scan, dtype=torch.uint8, device="cuda")
    Th = np.ascontiguousarray(guess)
    _lib.check(lib.vfm_icp_step_nearest_desc(src.data_ptr(), n_scan, Th.ctypes.data, moved.data_ptr(), sd.data_ptr(), sn.data_ptr(), sh.data_ptr(), f,
                                             g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(), g.desc.data_ptr(), g.norm.data_ptr(),
                                             g.has.data_ptr(), g.n_voxels, g.voxel_size, 3 * sigma, tgt.data_ptr(), valid.data_ptr(), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(tgt.cpu().numpy(), hist[0][2])
    np.testing.assert_array_equal(valid.cpu().numpy(), hist[0][3])
    # the weight changes the answer: the plain 3-D search from the same positions picks other neighbours for some points
    plain_t = torch.empty_like(src)
    plain_v = torch.empty_like(valid)
    _lib.check(lib.vfm_icp_nearest(moved.data_ptr(), n_scan, g.keys.data_ptr(), g.start.data_ptr(), g.pts.data_ptr(), g.n_voxels, g.voxel_size,
                                   3 * sigma, plain_t.data_ptr(), plain_v.data_ptr(), None))
    torch.cuda.synchronize()
    if f > 1:
        assert (plain_t != tgt).any(dim=1).sum().item() > 0
    # rows without descriptors behave as the 3-D search does
    if zero_rows:
        z = np.flatnonzero(~(scan[:, 3:] != 0).any(1))
        np.testing.assert_array_equal(tgt.cpu().numpy()[z], plain_t.cpu().numpy()[z])
    # an empty descriptor map hands the guess back (Registration.cpp:389)
    np.testing.assert_array_equal(register_frame(scan, get_voxel_hash_map(cfg), guess, 6.0, 0.6), guess)
