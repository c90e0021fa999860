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


def test_error_table_matches_reference_print_errors(golden):
    """Row F4: recall (registration_node.py:1021-1025 == print_errors.py:8-13) and the paper-table rows of
    print_errors.main (print_errors.py:27-56) against OUTPUTS OF THE REFERENCE's print_errors.py
    (tests/golden/make_golden.py::gen_print_errors)."""
    import numpy as np
    from oracle import oracle as orc
    from vfmreg.evaluation import Evaluation
    g = golden("print_errors.npz")
    ev = Evaluation()
    for m, rot, trans in zip(g["methods"], g["rot"], g["trans"]):
        ev.rot_errors[str(m)] = list(rot)
        ev.trans_errors[str(m)] = list(trans)
    for i, m in enumerate(g["methods"]):
        for j, (t, r) in enumerate(g["thresholds"]):
            assert ev.compute_success_rate(str(m), t, r) == g["rates"][i, j]
            assert orc.success_rate(g["trans"][i], g["rot"][i], t, r) == g["rates"][i, j]
    assert ev.error_string() == str(g["error_txt"])
    assert "vfm_ransac_icp" in ev.summary()
