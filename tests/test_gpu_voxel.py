"""Row F1 on the GPU: voxel down-sampling / voxel-hash-map insertion (csrc/voxel.hip) against the oracle
(sort-based restatement of Preprocessing.cpp:50-137 and VoxelHashMap.cpp:733-770)."""
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


def test_voxel_down_sample_mirror():
    from oracle import oracle as orc
    from vfmreg.voxelization import voxel_down_sample
    rng = np.random.default_rng(0)
    pts = rng.uniform(-30, 30, (5000, 7))
    for vs in (0.25, 1.0, 5.0):
        out = voxel_down_sample(pts, vs)                                   # voxelization.py:27
        np.testing.assert_array_equal(out, pts[orc.voxel_first(pts, vs, 1)])
        v = np.trunc(out[:, :3] / vs).astype(int)
        assert len(np.unique(v, axis=0)) == len(v)
    # truncation toward zero (Preprocessing.cpp:58): -0.3 and 0.3 share voxel 0
    two = np.array([[-0.3, 0.1, 0.1], [0.3, 0.1, 0.1], [1.2, 0.1, 0.1]])
    assert len(voxel_down_sample(two, 1.0)) == 2
    assert voxel_down_sample(np.zeros((0, 3)), 1.0).shape == (0, 3)
    with pytest.raises(ValueError, match="Invalid shape"):
        voxel_down_sample(np.zeros((4, 2)), 1.0)


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
    np.testing.assert_array_equal(m.point_cloud(), pts[orc.voxel_first(pts, 1.0, 20)])
    assert m.empty_n() and not m.empty()
    with pytest.raises(ValueError, match="Invalid shape"):
        m.add_points(np.zeros((3, 2)))
    m.clear()
    assert m.empty()
