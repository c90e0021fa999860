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


def test_cu_mask_words_for_any_unit_count():
    """the compute-unit mask of pipeline.masked_stream on parts whose unit count is not a multiple of 32 (the words used
    to be total // 32, and the bit index was taken before the wrap)."""
    from vfmreg.pipeline import cu_mask_words
    assert cu_mask_words(32, 0, 256) == [0xFFFFFFFF, 0, 0, 0, 0, 0, 0, 0]
    assert cu_mask_words(8, 28, 256)[:2] == [0xF0000000, 0x0000000F]
    for total in (256, 304, 228, 110, 64, 33):
        for ncu, off in ((1, 0), (total, 0), (total // 3, total - 5), (total - 1, 7)):
            w = cu_mask_words(ncu, off, total)
            assert len(w) == (total + 31) // 32
            bits = [j for j in range(32 * len(w)) if (w[j // 32] >> (j % 32)) & 1]
            assert bits == sorted({i % total for i in range(off, off + ncu)}) and all(b < total for b in bits)
            assert all(0 <= x < 2 ** 32 for x in w)
    import pytest
    with pytest.raises(ValueError):
        cu_mask_words(0, 0, 256)
    with pytest.raises(ValueError):
        cu_mask_words(300, 0, 256)


def test_icp_solve_of_a_singular_system_does_not_raise():
    """one or two (or collinear) surviving pairs make the 6 x 6 normal equations singular; the reference's LDLT does not
    throw -- the loops stop instead of letting numpy's LinAlgError escape."""
    from vfmreg.icp import _solve6
    A = np.diag([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    b = np.arange(6.0)
    np.testing.assert_array_equal(_solve6(A, b), np.linalg.solve(A, b))
    assert _solve6(np.zeros((6, 6)), b) is None
    J = np.outer(np.arange(1.0, 7.0), np.arange(1.0, 7.0))   # rank 1
    assert _solve6(J, b) is None or np.isfinite(_solve6(J, b)).all()
