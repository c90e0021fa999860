"""Row F1 on the GPU: voxel down-sampling / voxel-hash-map insertion (csrc/voxel.hip) against the oracle:
the survivor set (Preprocessing.cpp:50-137, VoxelHashMap.cpp:733-770) AND the reference's emission order
(tsl::robin_map iteration order, Preprocessing.cpp:64-69 / VoxelHashMap.cpp:628-676; oracle:
``orc_voxel_robin``), including the chained voxelisations of registration_node.py:399-414 and 556-580."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("n,extent,vs,K", [(5000, 30.0, 1.0, 1), (5000, 30.0, 0.25, 1), (20000, 4.0, 1.0, 20),
                                           (20000, 4.0, 1.0, 3), (200000, 60.0, 1.0, 20), (1, 1.0, 1.0, 1)])
def test_voxel_first_matches_oracle(n, extent, vs, K):
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(n + K)
    pts = np.c_[rng.uniform(-extent, extent, (n, 3)), rng.standard_normal((n, 4))]
    pts[n // 2] = pts[0]                       # exact duplicate point
    keep = ops.voxel_first(torch.from_numpy(pts).cuda(), vs, K).cpu().numpy()
    np.testing.assert_array_equal(keep, orc.voxel_first(pts, vs, K))


def test_voxel_keys_do_not_alias_at_large_coordinates():
    """Absolute / UTM coordinates: |xyz / voxel_size| beyond 2^20 (the old packed 21-bit key aliased there)."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(5)
    base = np.array([3.2e5, 4.7e6, 120.0])
    pts = base + rng.uniform(-20, 20, (20000, 3))
    for vs in (0.1, 0.25):
        assert np.abs(pts / vs).max() > 2 ** 20
        keep = ops.voxel_first(torch.from_numpy(pts).cuda(), vs, 1).cpu().numpy()
        np.testing.assert_array_equal(keep, orc.voxel_first(pts, vs, 1))
        got = ops.voxel_robin(torch.from_numpy(pts).cuda(), vs).cpu().numpy()
        np.testing.assert_array_equal(got, orc.voxel_robin(pts, vs))


@pytest.mark.parametrize("n,extent,vs", [(1, 1.0, 1.0), (7, 3.0, 1.0), (700, 10.0, 0.5), (5000, 30.0, 1.0), (5000, 30.0, 0.25),
                                         (120000, 60.0, 0.25), (400000, 80.0, 0.5)])
def test_downsample_order_is_the_containers(n, extent, vs):
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(n)
    pts = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.15]
    got, info = ops.voxel_robin(torch.from_numpy(pts).cuda(), vs, return_info=True)
    ref, rinfo = orc.voxel_robin(pts, vs, return_info=True)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert info[0] == rinfo[0] and info[1] == rinfo[1]        # bucket count, voxel count


def test_downsample_order_with_wrapping_clusters():
    """Voxels whose home buckets are the last ones of the table: the cluster wraps to bucket 0."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(11)
    wrapped = 0
    for trial in range(12):
        n = int(rng.integers(50, 3000))
        c = int(np.ceil(np.float32(n) / np.float32(0.5)))
        B = 1 << int(c - 1).bit_length()
        cand = rng.integers(-300, 300, (200000, 3)).astype(np.int32)
        h = orc.voxel_hash(cand, orc.HASH_MUL_DOWNSAMPLE) & (B - 1)
        near, low = cand[h >= B - 3][:int(rng.integers(4, 12))], cand[h <= 2][:int(rng.integers(0, 6))]
        rest = cand[rng.integers(0, len(cand), n - len(near) - len(low))]
        vox = np.concatenate([near, low, rest])
        vox = vox[rng.permutation(len(vox))]
        pts = (vox.astype(np.float64) + np.where(vox >= 0, 0.5, -0.5)) * 0.5
        got, info = ops.voxel_robin(torch.from_numpy(pts).cuda(), 0.5, return_info=True)
        np.testing.assert_array_equal(got.cpu().numpy(), orc.voxel_robin(pts, 0.5))
        wrapped += int(info[3] > 0)
    assert wrapped >= 6


@pytest.mark.parametrize("n,extent,K", [(3, 2.0, 20), (300, 6.0, 20), (513, 30.0, 20), (8000, 3.0, 20), (8000, 40.0, 2),
                                        (200000, 60.0, 20)])
def test_growing_map_order_is_the_containers(n, extent, K):
    """A default-constructed VoxelHashMap that doubles as it fills (VoxelHashMap.hpp:117, AddPoints)."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(n + K)
    pts = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.15]
    got, info = ops.voxel_robin(torch.from_numpy(pts).cuda(), 1.0, K, reserve=False, hash_mul=ops.HASH_MAP, return_info=True)
    ref, rinfo = orc.voxel_robin(pts, 1.0, K, False, orc.HASH_MUL_MAP, return_info=True)
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    assert info[0] == rinfo[0] and info[1] == rinfo[1]


def test_voxel_down_sample_mirror():
    from oracle import oracle as orc
    from vfmreg.voxelization import voxel_down_sample
    rng = np.random.default_rng(0)
    pts = rng.uniform(-30, 30, (5000, 7))
    for vs in (0.25, 1.0, 5.0):
        out = voxel_down_sample(pts, vs)                                   # voxelization.py:27
        np.testing.assert_array_equal(out, orc.voxel_down_sample(pts, vs))
        np.testing.assert_array_equal(np.sort(out.view(np.dtype((np.void, 56))).ravel()),
                                      np.sort(pts[orc.voxel_first(pts, vs, 1)].view(np.dtype((np.void, 56))).ravel()))
        v = np.trunc(out[:, :3] / vs).astype(int)
        assert len(np.unique(v, axis=0)) == len(v)
    # truncation toward zero (Preprocessing.cpp:58): -0.3 and 0.3 share voxel 0
    two = np.array([[-0.3, 0.1, 0.1], [0.3, 0.1, 0.1], [1.2, 0.1, 0.1]])
    assert len(voxel_down_sample(two, 1.0)) == 2
    assert voxel_down_sample(np.zeros((0, 3)), 1.0).shape == (0, 3)
    with pytest.raises(ValueError, match="Invalid shape"):
        voxel_down_sample(np.zeros((4, 2)), 1.0)


def test_chained_voxelisation_equals_reference_chain():
    """registration_node.py:399-414 (0.5 -> 1.0 -> [transform] -> 5.0 m, retry 1.0) and 556-580 (0.25 m per cloud,
    then 0.25 m on the concatenation): every level keeps the first point per voxel of the previous level's
    CONTAINER order, so the chain only matches if each level's order does."""
    from oracle import oracle as orc
    from vfmreg.voxelization import voxel_down_sample
    rng = np.random.default_rng(3)
    scan = np.c_[rng.uniform(-50, 50, (60000, 3)) * [1, 1, 0.1], rng.standard_normal((60000, 5))]
    a = voxel_down_sample(voxel_down_sample(scan, 0.5), 1.0)
    b = orc.voxel_down_sample(orc.voxel_down_sample(scan, 0.5), 1.0)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(voxel_down_sample(a, 5.0), orc.voxel_down_sample(b, 5.0))
    np.testing.assert_array_equal(voxel_down_sample(a, 1.0), orc.voxel_down_sample(b, 1.0))
    # and the input-order chain (what round 1 shipped) is a different point set
    c = scan
    for vs in (0.5, 1.0, 5.0):
        c = c[orc.voxel_first(c, vs, 1)]
    assert set(map(tuple, c[:, :3])) != set(map(tuple, voxel_down_sample(a, 5.0)[:, :3]))
    # map accumulation: per-cloud 0.25 m, concatenate, 0.25 m again
    clouds = [rng.uniform(-30, 30, (20000, 3)) * [1, 1, 0.1] + [5.0 * k, 0, 0] for k in range(4)]
    got = voxel_down_sample(np.concatenate([voxel_down_sample(c, 0.25) for c in clouds]), 0.25)
    ref = orc.voxel_down_sample(np.concatenate([orc.voxel_down_sample(c, 0.25) for c in clouds]), 0.25)
    np.testing.assert_array_equal(got, ref)


def test_voxel_hash_map_caps_points_per_voxel():
    from oracle import oracle as orc
    from vfmreg.config import load_config
    from vfmreg.mapping import get_voxel_hash_map
    cfg = load_config(None, None)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-3, 3, (8000, 3))      # ~37 points per 1 m voxel -> the cap of 20 bites
    m = get_voxel_hash_map(cfg)
    m.add_points(pts[:5000])
    m.add_points(pts[5000:])                 # incremental insertion honours the earlier counts
    np.testing.assert_array_equal(m.point_cloud(), pts[orc.voxel_hash_map_points(pts, 1.0, 20)])
    np.testing.assert_array_equal(np.sort(orc.voxel_hash_map_points(pts, 1.0, 20)), orc.voxel_first(pts, 1.0, 20))
    assert m.empty_n() and not m.empty()
    with pytest.raises(ValueError, match="Invalid shape"):
        m.add_points(np.zeros((3, 2)))
    m.clear()
    assert m.empty()
    # 387-column points live in map_n_; point_cloud() falls back to their heads (VoxelHashMap.cpp:641-649)
    wide = np.c_[rng.uniform(-30, 30, (3000, 3)), rng.standard_normal((3000, 384))]
    m.add_points(wide)
    order = orc.voxel_hash_map_points(wide, 1.0, 20)
    np.testing.assert_array_equal(m.point_cloud_n(), wide[order])
    np.testing.assert_array_equal(m.point_cloud(), wide[order][:, :3])
    assert m.empty() and not m.empty_n()


def test_one_launch_downsample_equals_the_general_path_and_the_oracle():
    """voxel_robin_grid_kernel (round 5: VoxelDownsample of up to 2^18 points in one launch -- up to 256 resident workgroups walk through
    first-point table, counting sort by home bucket, clusters and the per-cluster replay together, grid-wide barriers between the
    phases) against the general multi-launch path and the oracle's robin-map replay: sizes on both sides of a workgroup's worth of
    points, dense and sparse tables, duplicates, the chained voxelisations of registration_node.py:399-414, tables whose last cluster
    wraps (clusters beyond the kernel's limit fall back to the general path)."""
    from oracle import oracle as orc
    from vfmreg import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(77)
    cases = [(1, 1.0, 1.0), (2, 1.0, 1.0), (63, 2.0, 0.5), (1024, 6.0, 0.5), (1025, 6.0, 1.0), (4097, 3.0, 1.0), (20000, 60.0, 0.5),
             (20000, 60.0, 5.0), (60000, 50.0, 0.5), (60000, 50.0, 0.05), (131072, 60.0, 0.25), (262144, 80.0, 0.5)]
    try:
        for n, extent, vs in cases:
            pts = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.15]
            if n > 10:
                pts[n // 2] = pts[0]
                pts[n - 1] = pts[1]
            ref, rinfo = orc.voxel_robin(pts, vs, return_info=True)
            d = torch.from_numpy(pts).cuda()
            lib.vfm_debug_set_voxel_small(0)
            gen, ginfo = ops.voxel_robin(d, vs, return_info=True)
            lib.vfm_debug_set_voxel_small(1)
            one, oinfo = ops.voxel_robin(d, vs, return_info=True)
            np.testing.assert_array_equal(one.cpu().numpy(), ref, err_msg=str((n, extent, vs)))
            np.testing.assert_array_equal(gen.cpu().numpy(), ref, err_msg=str((n, extent, vs)))
            assert oinfo[:2] == ginfo[:2] == list(rinfo[:2]), (n, oinfo, ginfo, rinfo)
        # chained: .5 -> 1.0 -> 5.0 on the survivors, as the node does
        pts = rng.uniform(-50, 50, (60000, 3)) * [1, 1, 0.1]
        cur_h, cur_d = pts, torch.from_numpy(pts).cuda()
        lib.vfm_debug_set_voxel_small(1)
        for vs in (0.5, 1.0, 5.0):
            o = ops.voxel_robin(cur_d, vs)
            r = orc.voxel_robin(cur_h, vs)
            np.testing.assert_array_equal(o.cpu().numpy(), r)
            cur_h, cur_d = cur_h[r], cur_d[o]
        # wrapping clusters (homes in the last buckets): the construction of test_downsample_order_with_wrapping_clusters
        wrapped = 0
        for trial in range(12):
            n = int(rng.integers(50, 3000))
            c = int(np.ceil(np.float32(n) / np.float32(0.5)))
            B = 1
            while B < c:
                B <<= 1
            xs = []
            x = 0
            while len(xs) < n:       # voxels (x, 0, 0) whose VoxelHash & (B - 1) falls into the last 8 buckets or anywhere
                h = ((x * 73856093) & 0xFFFFFFFF) & ((1 << 20) - 1) & (B - 1)
                if h >= B - 8 or rng.random() < 0.3:
                    xs.append(x)
                x += 1
            pts = np.c_[np.array(xs, dtype=np.float64) + 0.5, np.full(n, 0.5), np.full(n, 0.5)]
            pts = pts[rng.permutation(n)]
            ref, rinfo = orc.voxel_robin(pts, 1.0, return_info=True)
            lib.vfm_debug_set_voxel_small(1)
            one, oinfo = ops.voxel_robin(torch.from_numpy(pts).cuda(), 1.0, return_info=True)
            np.testing.assert_array_equal(one.cpu().numpy(), ref, err_msg=f"trial {trial}")
            wrapped += int(oinfo[3] > 0)
        assert wrapped >= 3, wrapped
    finally:
        lib.vfm_debug_set_voxel_small(1)


def test_chained_levels_without_a_read_back_equal_the_level_by_level_path():
    """vfm_voxel_robin_level (round 5): the three chained VoxelDownsample()s of registration_node.py:399-414 enqueued as three launches --
    each level takes the survivors of the one before as rows of the raw cloud, its size read on the device, the third level on the points
    moved by a pose -- against the level-by-level path (gather, transform, vfm_voxel_robin, read-back) and the oracle."""
    from oracle import oracle as orc
    from vfmreg import ops, synth
    rng = np.random.default_rng(12)
    for n, extent in ((5, 1.0), (3000, 20.0), (20000, 60.0), (60000, 50.0), (200000, 80.0)):
        pts = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.15]
        pose = synth.random_pose(rng)
        d = torch.from_numpy(pts).cuda()
        T = torch.from_numpy(np.ascontiguousarray(pose)).cuda()
        l1 = ops.voxel_robin_level(d, 0.5)
        l2 = ops.voxel_robin_level(d, 1.0, idx=l1["keep"], n_dev=l1["count"], n_max=n)
        l3 = ops.voxel_robin_level(d, 5.0, idx=l2["keep"], n_dev=l2["count"], n_max=n, T=T, want_local=True)
        torch.cuda.synchronize()
        infos = [l["info"].cpu().numpy() for l in (l1, l2, l3)]
        assert all(i[5] == 1 and i[1] > 0 for i in infos), infos
        n1, n2, n3 = (int(l["count"].item()) for l in (l1, l2, l3))
        assert [n1, n2, n3] == [int(i[1]) for i in infos]
        # level by level on the device
        o1 = ops.voxel_robin(d, 0.5)
        x1 = d[o1]
        o2 = ops.voxel_robin(x1, 1.0)
        x2 = x1[o2]
        o3 = ops.voxel_robin(ops.transform_xyz(x2, T), 5.0)
        np.testing.assert_array_equal(l1["keep"][:n1].cpu().numpy(), o1.cpu().numpy())
        np.testing.assert_array_equal(l2["keep"][:n2].cpu().numpy(), o1[o2].cpu().numpy())
        np.testing.assert_array_equal(l3["local"][:n3].cpu().numpy(), o3.cpu().numpy())
        np.testing.assert_array_equal(l3["keep"][:n3].cpu().numpy(), o1[o2][o3].cpu().numpy())
        # and the oracle
        r1 = orc.voxel_robin(pts, 0.5)
        r2 = orc.voxel_robin(pts[r1], 1.0)
        moved = ops.transform_xyz(x2, T).cpu().numpy()
        r3 = orc.voxel_robin(moved, 5.0)
        np.testing.assert_array_equal(o1.cpu().numpy(), r1)
        np.testing.assert_array_equal(o2.cpu().numpy(), r2)
        np.testing.assert_array_equal(o3.cpu().numpy(), r3)


def test_one_launch_downsample_from_several_threads_beside_other_work():
    """The one-launch kernel's grid-wide barriers need all of its workgroups resident: four host threads, each on a stream of its own,
    down-sample clouds of different sizes at once while a fifth stream keeps the device busy with large products; every result equals
    the oracle's (a grid that could not become resident would raise the kernel's abort flag and take the general path -- still the
    same order)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(5)
    clouds = [rng.uniform(-60, 60, (n, 3)) * [1, 1, 0.15] for n in (20000, 60000, 1700, 131072)]
    refs = [orc.voxel_robin(c, 0.5) for c in clouds]
    dev = [torch.from_numpy(c).cuda() for c in clouds]
    busy = torch.randn(8192, 8192, device="cuda")
    stop = []

    def load():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop:
                (busy @ busy).sum().item()

    def work(k):
        st = torch.cuda.Stream()
        outs = []
        with torch.cuda.stream(st):
            for _ in range(15):
                outs.append(ops.voxel_robin(dev[k], 0.5).cpu().numpy())
        return outs

    torch.cuda.synchronize()
    with ThreadPoolExecutor(5) as ex:
        bg = ex.submit(load)
        futs = [ex.submit(work, k) for k in range(4)]
        res = [f.result() for f in futs]
        stop.append(1)
        bg.result()
    for k in range(4):
        for o in res[k]:
            np.testing.assert_array_equal(o, refs[k])


def test_replay_in_the_lds_equals_the_replay_behind_a_radix_sort():
    """robin_replay2_kernel (round 5: a cluster's members ordered by the replaying thread itself, the replay in a per-thread slice of the
    LDS; clusters beyond 24 entries in place in global memory) against round 4's form (radix sort by (cluster, arrival), replay through
    global memory) and the oracle: down-sampling and growing maps, a dense table with long clusters (more voxels than 20-bit hashes spread),
    wrapping clusters."""
    from oracle import oracle as orc
    from vfmreg import _lib, ops
    lib = _lib.load()
    rng = np.random.default_rng(91)
    try:
        lib.vfm_debug_set_voxel_small(0)   # (the general path: the one-launch kernel would take the down-sampling cases)
        for n, extent, vs, K, reserve in [(700, 10.0, 0.5, 1, True), (20000, 60.0, 0.5, 1, True), (120000, 60.0, 0.25, 1, True),
                                          (8000, 3.0, 1.0, 20, False), (200000, 60.0, 1.0, 20, False), (400000, 80.0, 0.5, 1, True)]:
            pts = rng.uniform(-extent, extent, (n, 3)) * [1, 1, 0.15]
            d = torch.from_numpy(pts).cuda()
            hm = ops.HASH_DOWNSAMPLE if reserve else ops.HASH_MAP
            ref = orc.voxel_robin(pts, vs, K, reserve, orc.HASH_MUL_DOWNSAMPLE if reserve else orc.HASH_MUL_MAP)
            outs = []
            for mode in (2, 3):
                lib.vfm_debug_set_voxel_small(mode)
                outs.append(ops.voxel_robin(d, vs, K, reserve=reserve, hash_mul=hm).cpu().numpy())
            np.testing.assert_array_equal(outs[0], ref, err_msg=str((n, K, "radix sort + global replay")))
            np.testing.assert_array_equal(outs[1], ref, err_msg=str((n, K, "replay in the LDS")))
    finally:
        lib.vfm_debug_set_voxel_small(3)
        lib.vfm_debug_set_voxel_small(1)


def _cloud_with_a_long_cluster(n, rng, members=400):
    """voxels (x, 0, 0) of which `members` have their home bucket in one window of 100 buckets of the reserved table: one run of
    occupied buckets longer than the one-launch kernel's limit (GRID_MAX_CLUSTER = 192)"""
    c = int(np.ceil(np.float32(n) / np.float32(0.5)))
    B = 1
    while B < c:
        B <<= 1
    xs, dense, x = [], 0, 0
    while len(xs) < n:
        h = ((x * 73856093) & 0xFFFFFFFF) & ((1 << 20) - 1) & (B - 1)
        if 1000 <= h < 1100 and dense < members:
            xs.append(x)
            dense += 1
        elif not (900 <= h < 1700) and rng.random() < 0.2 and len(xs) < n - (members - dense):
            xs.append(x)
        x += 1
    pts = np.c_[np.array(xs, dtype=np.float64) + 0.5, np.full(n, 0.5), np.full(n, 0.5)]
    return pts[rng.permutation(n)]


def test_a_level_the_kernel_cannot_reproduce_leaves_an_empty_level_to_the_chain():
    """vfm_voxel_robin_level on a cloud with a cluster beyond the one-launch kernel's limit reports info[1] = -1 -- and used
    to publish count = nv with keep_out unwritten, so the next levels of a chain (enqueued without a read-back) gathered points through
    uninitialised indices.  Now such a level publishes count 0: the later levels are empty, the caller redoes the chain level by level
    (RegistrationNode._voxel_chain returns None), and vfm_voxel_robin on the same cloud takes the general path: the oracle's order."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(31)
    n = 3000
    pts = _cloud_with_a_long_cluster(n, rng)
    d = torch.from_numpy(pts).cuda()
    for _ in range(3):
        l1 = ops.voxel_robin_level(d, 1.0)
        l1["keep"].fill_(1 << 40)           # what uninitialised memory may hold
        l2 = ops.voxel_robin_level(d, 2.0, idx=l1["keep"], n_dev=l1["count"], n_max=n)
        l3 = ops.voxel_robin_level(d, 5.0, idx=l2["keep"], n_dev=l2["count"], n_max=n, want_local=True)
        torch.cuda.synchronize()
        i1, i2, i3 = (l["info"].cpu().numpy() for l in (l1, l2, l3))
        assert i1[5] == 1 and i1[1] == -1, i1
        assert [int(l["count"].item()) for l in (l1, l2, l3)] == [0, 0, 0]
        assert i2[5] == 1 and i3[5] == 1 and i2[1] == 0 and i3[1] == 0, (i2, i3)
    ref = orc.voxel_robin(pts, 1.0)
    np.testing.assert_array_equal(ops.voxel_robin(d, 1.0).cpu().numpy(), ref)


def test_one_launch_downsample_on_a_stream_with_few_compute_units():
    """the one-launch kernel's grid is clamped to what can be resident on the compute units the STREAM may use (occupancy x
    the stream's compute-unit mask): on a stream restricted to 8 units the 256-workgroup grid of round 5 could never pass its first
    barrier and every call span for seconds before it fell back.  Same order as the oracle's, and quickly."""
    import time
    from oracle import oracle as orc
    from vfmreg import ops
    from vfmreg.pipeline import masked_stream
    rng = np.random.default_rng(8)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    st = masked_stream(8, 16, ncu)
    for n in (1700, 20000, 60000, 200000):
        pts = rng.uniform(-60, 60, (n, 3)) * [1, 1, 0.15]
        ref = orc.voxel_robin(pts, 0.5)
        d = torch.from_numpy(pts).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            ops.voxel_robin(d, 0.5)
            t0 = time.perf_counter()
            got = ops.voxel_robin(d, 0.5)
            st.synchronize()
            dt = time.perf_counter() - t0
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        assert dt < 0.25, (n, dt)     # (a grid that is not resident spins for seconds)
