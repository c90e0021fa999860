"""Row F1, container order: the oracle's bucket-level restatement of tsl::robin_map (vfm_oracle.c,
``orc_voxel_robin``: insert_impl / insert_value_impl / rehash_impl / reserve of Tessil robin-map v1.2.1,
VoxelHash of Preprocessing.cpp:41-46 and VoxelHashMap.hpp:72-77) against
  * the survivor SET (first K points per voxel, sort-based ``orc_voxel_first``),
  * a structurally different derivation of the ORDER (``robin_order_by_clusters``: stable sort by home
    bucket -> clusters -> per-cluster replay), which is the decomposition csrc/voxel.hip runs on the GPU,
  * hand-worked tables.
tsl::robin_map is not in /root/reference (fetched by 3rdparty/tsl_robin/tsl_robin.cmake:24) nor in the image:
parity with the real header is UNPINNED; these tests pin the two restatements to each other."""
import numpy as np

from oracle import oracle as orc


def _voxel_sequence(idx, vox, first):
    rank = {tuple(k): i for i, k in enumerate(vox[first])}
    out, last = [], None
    for p in idx:
        r = rank[tuple(vox[p])]
        if r != last:
            out.append(r)
            last = r
    return np.array(out, dtype=np.int64)


def test_hand_worked_tables():
    # VoxelHash((x,0,0)) = (x * 73856093) & 0xFFFFF.  Reserve(4) -> 8 buckets.
    h = lambda x: (x * 73856093) & 0xFFFFF
    xs = [x for x in range(1, 4000) if h(x) & 7 == 3][:3] + [x for x in range(1, 4000) if h(x) & 7 == 4][:1]
    a, b, c, e = xs                     # a, b, c share home bucket 3; e lives in 4
    # arrivals a, e, b, c:  a->3, e->4, b: passes a (equal distance), takes 4 from e (e -> 5),
    # c: passes a and b, takes 5 from e (e -> 6)  =>  a b c e
    pts = np.array([[a, 0, 0], [e, 0, 0], [b, 0, 0], [c, 0, 0]], dtype=np.float64) + 0.5
    idx, info = orc.voxel_robin(pts, 1.0, 1, True, orc.HASH_MUL_DOWNSAMPLE, return_info=True)
    assert info[0] == 8 and list(idx) == [0, 2, 3, 1]
    # leapfrog: arrivals e, a, then b, then f (home 2): a displaced entry jumps over its equal-distance peers.
    f = [x for x in range(1, 4000) if h(x) & 7 == 2][0]
    g = [x for x in range(1, 4000) if h(x) & 7 == 2][1]
    # a->3, b->4, f->2, g: home 2, passes f, takes 3 from a; a (d=1) meets b (d=1): no swap, a goes to 5
    pts = np.array([[a, 0, 0], [b, 0, 0], [f, 0, 0], [g, 0, 0]], dtype=np.float64) + 0.5
    idx = orc.voxel_robin(pts, 1.0, 1, True, orc.HASH_MUL_DOWNSAMPLE)
    assert list(idx) == [2, 3, 1, 0]    # f g b a : the later b now precedes the earlier a
    # a growing map: 0 -> 2 -> 4 -> 8 buckets (rehash when size() >= buckets/2)
    idx, info = orc.voxel_robin(pts, 1.0, 20, False, orc.HASH_MUL_MAP, return_info=True)
    assert info[0] == 8 and info[3] == 3 and sorted(idx) == [0, 1, 2, 3]
    # empty input
    assert len(orc.voxel_robin(np.zeros((0, 3)), 1.0)) == 0


def test_two_derivations_agree_random():
    rng = np.random.default_rng(0)
    for trial in range(60):
        n = int(rng.integers(0, 3000)) if trial < 54 else int(rng.integers(20000, 60000))
        span = float(rng.choice([2, 5, 20, 100]))
        pts = rng.uniform(-span, span, (n, 3))
        vs = float(rng.choice([0.1, 0.5, 1.0, 5.0]))
        vox = (pts / vs).astype(np.int32)
        first = orc.voxel_first(pts, vs, 1)
        for reserve, K, mul in ((True, 1, orc.HASH_MUL_DOWNSAMPLE), (False, 1, orc.HASH_MUL_MAP),
                                (False, 3, orc.HASH_MUL_MAP)):
            idx, info = orc.voxel_robin(pts, vs, K, reserve, mul, return_info=True)
            np.testing.assert_array_equal(np.sort(idx), orc.voxel_first(pts, vs, K))     # the SET
            assert info[1] == len(first)
            o = orc.robin_order_by_clusters(orc.voxel_hash(vox[first], mul), n if reserve else None)
            np.testing.assert_array_equal(_voxel_sequence(idx, vox, first), o)            # the ORDER
            if K > 1:   # inside a voxel block: insertion order
                pos = {int(p): i for i, p in enumerate(idx)}
                key = [tuple(v) for v in vox[idx]]
                for a, b in zip(range(len(idx) - 1), range(1, len(idx))):
                    if key[a] == key[b]:
                        assert idx[a] < idx[b]


def test_two_derivations_agree_when_entries_wrap():
    """Homes in the last buckets of the table: the cluster wraps to bucket 0 and iteration starts with it."""
    rng = np.random.default_rng(1)
    wrapped = 0
    for trial in range(20):
        n = int(rng.integers(50, 2000))
        c = int(np.ceil(np.float32(n) / np.float32(0.5)))
        B = 1 << int(c - 1).bit_length()
        cand = rng.integers(-300, 300, (200000, 3)).astype(np.int32)
        h = orc.voxel_hash(cand, orc.HASH_MUL_DOWNSAMPLE) & (B - 1)
        near, low = cand[h >= B - 3][:int(rng.integers(4, 12))], cand[h <= 2][:int(rng.integers(0, 6))]
        rest = cand[rng.integers(0, len(cand), n - len(near) - len(low))]
        vox = np.concatenate([near, low, rest])
        vox = vox[rng.permutation(len(vox))]
        pts = (vox.astype(np.float64) + np.where(vox >= 0, 0.5, -0.5)) * 0.5
        idx, info = orc.voxel_robin(pts, 0.5, 1, True, orc.HASH_MUL_DOWNSAMPLE, return_info=True)
        assert info[0] == B
        v2 = (pts / 0.5).astype(np.int32)
        first = orc.voxel_first(pts, 0.5, 1)
        hh = orc.voxel_hash(v2[first], orc.HASH_MUL_DOWNSAMPLE)
        sv = _voxel_sequence(idx, v2, first)
        np.testing.assert_array_equal(sv, orc.robin_order_by_clusters(hh, n))
        wrapped += int((hh[sv[0]] & (B - 1)) > B // 2)
    assert wrapped >= 10


def test_chained_downsampling_depends_on_order():
    """registration_node.py:399-414: 0.5 m -> 1.0 m -> 5.0 m.  The survivors of the chain in container
    order differ from those of an input-order chain -- the reason the order is reproduced at all."""
    rng = np.random.default_rng(2)
    pts = rng.uniform(-40, 40, (30000, 3)) * [1, 1, 0.1]
    a = orc.voxel_down_sample(orc.voxel_down_sample(orc.voxel_down_sample(pts, 0.5), 1.0), 5.0)
    b = pts
    for vs in (0.5, 1.0, 5.0):
        b = b[orc.voxel_first(b, vs, 1)]
    assert len(a) == len(b)                                   # same voxels ...
    va, vb = np.trunc(a / 5.0).astype(int), np.trunc(b / 5.0).astype(int)
    assert set(map(tuple, va)) == set(map(tuple, vb))
    assert set(map(tuple, a)) != set(map(tuple, b))           # ... but other representatives
