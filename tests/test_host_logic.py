"""CPU tests of the host-side logic that needs no GPU: voxel down-sampling / voxel hash map
bookkeeping (rows F1 / A5 container), config constants, error conventions."""
import numpy as np
import pytest

from oracle import oracle as orc
from vfmreg import voxelization as VX
from vfmreg.config import load_config


def test_voxel_down_sample_first_point_per_voxel():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-30, 30, (5000, 7))
    for vs in (0.25, 1.0, 5.0):
        keep = VX.first_per_voxel(pts, vs, 1)
        np.testing.assert_array_equal(keep, orc.voxel_first(pts, vs, 1))
        out = VX.voxel_down_sample(pts, vs)
        np.testing.assert_array_equal(out, pts[keep])
        # one point per voxel, truncation toward zero (Preprocessing.cpp:58): (-0.3, 0.3) share voxel 0
        v = np.trunc(out[:, :3] / vs).astype(int)
        assert len(np.unique(v, axis=0)) == len(v)
    two = np.array([[-0.3, 0.1, 0.1], [0.3, 0.1, 0.1], [1.2, 0.1, 0.1]])
    assert len(VX.voxel_down_sample(two, 1.0)) == 2
    with pytest.raises(ValueError, match="Invalid shape"):
        VX.voxel_down_sample(np.zeros((4, 2)), 1.0)


def test_voxel_hash_map_caps_points_per_voxel():
    from vfmreg.mapping import VoxelHashMap, get_voxel_hash_map
    cfg = load_config(None, None)
    assert (cfg.mapping.voxel_size, cfg.mapping.max_points_per_voxel, cfg.data.max_range,
            cfg.adaptive_threshold.initial_threshold) == (1.0, 20, 100.0, 2.0)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-3, 3, (8000, 3))  # ~37 points per 1 m voxel -> the cap of 20 bites
    m = get_voxel_hash_map(cfg)
    m.add_points(pts[:5000])
    m.add_points(pts[5000:])               # incremental insertion honours earlier counts
    ref = pts[orc.voxel_first(pts, 1.0, 20)]
    got = m.point_cloud()
    assert len(got) == len(ref)
    np.testing.assert_array_equal(np.sort(got.view("f8,f8,f8"), axis=0), np.sort(ref.view("f8,f8,f8"), axis=0))
    assert m.empty_n() and not m.empty()
    with pytest.raises(ValueError, match="Invalid shape"):
        m.add_points(np.zeros((3, 2)))
    m.clear()
    assert m.empty()


def test_synth_pair_statistics():
    from vfmreg import synth
    p = synth.make_pair(2000, 10000, 384, seed=42)
    inl = p["match"] >= 0
    cos = (p["q_desc"][inl] * p["b_desc"][p["match"][inl]]).sum(1)
    assert 0.85 < cos.mean() < 0.95 and 0.4 < inl.mean() < 0.6
    back = p["q_xyz"][inl] @ p["T_gt"][:3, :3].T + p["T_gt"][:3, 3]
    assert np.abs(back - p["b_xyz"][p["match"][inl]]).max() < 0.15
