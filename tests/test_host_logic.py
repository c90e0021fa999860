"""CPU tests of host-side logic that needs no GPU: config constants, the synthetic generator."""
import numpy as np
import pytest

from oracle import oracle as orc
from vfmreg.config import load_config


def test_config_constants():
    cfg = load_config(None, None)
    assert (cfg.mapping.voxel_size, cfg.mapping.max_points_per_voxel, cfg.data.max_range,
            cfg.adaptive_threshold.initial_threshold) == (1.0, 20, 100.0, 2.0)


def test_synth_pair_statistics():
    from vfmreg import synth
    p = synth.make_pair(2000, 10000, 384, seed=42)
    inl = p["match"] >= 0
    cos = (p["q_desc"][inl] * p["b_desc"][p["match"][inl]]).sum(1)
    assert 0.85 < cos.mean() < 0.95 and 0.4 < inl.mean() < 0.6
    back = p["q_xyz"][inl] @ p["T_gt"][:3, :3].T + p["T_gt"][:3, 3]
    assert np.abs(back - p["b_xyz"][p["match"][inl]]).max() < 0.15
