"""Near-duplicate-rich maps: real lifted descriptors are bilinear interpolations of a 16 x 21
patch grid (image_features.py:104-110, prepare_scenes.py:85-104), so neighbouring map points differ by less than the
fp16 coarse window.  The fp32 refinement (match_refine_kernel) must keep indices bit-equal to the fp64 oracle and keep
the all-pairs fallback for the pathological cases only."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def manifold(rng, rows, d, G, jitter=0.0):
    """rows on a 2-D bilinear manifold spanned by a G x G grid of anchor features (what A3's gather produces)"""
    A = rng.standard_normal((G, G, d)).astype(np.float32)
    xy = rng.uniform(0, G - 1 - 1e-3, (rows, 2))
    i, j = xy[:, 0].astype(int), xy[:, 1].astype(int)
    fx, fy = (xy[:, 0] - i)[:, None].astype(np.float32), (xy[:, 1] - j)[:, None].astype(np.float32)
    out = (A[i, j] * (1 - fx) * (1 - fy) + A[i + 1, j] * fx * (1 - fy) + A[i, j + 1] * (1 - fx) * fy + A[i + 1, j + 1] * fx * fy)
    if jitter:
        out = out + jitter * rng.standard_normal(out.shape).astype(np.float32)
    return np.ascontiguousarray(out, dtype=np.float32), xy


def _stats(ops, q, b):
    """run the split search on explicit buffers so that the counters of the workspace can be read"""
    from vfmreg import _lib
    lib = _lib.load()
    lib.vfm_debug_set_match_stats(1)
    Q, B = ops.PreparedRows(q), ops.PreparedRows(b)
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(Q.rows, B.rows, Q.d), dtype=torch.uint8, device="cuda")
    idx, sim = ops.match_search(Q, B, ws=ws)
    out = (C.c_int32 * 64)()
    _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), Q.rows, B.rows, C.cast(out, C.c_void_p)))
    lib.vfm_debug_set_match_stats(0)
    return idx, sim, list(out)


@pytest.mark.parametrize("n,m,G,jitter", [(2000, 20000, 12, 0.0), (1500, 30000, 6, 1e-4), (700, 9000, 3, 0.0)])
def test_manifold_map_equals_oracle(n, m, G, jitter):
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(n + G)
    b, _ = manifold(rng, m, 384, G, jitter)
    pick = rng.permutation(m)[:n]
    q = b[pick] + 0.05 * rng.standard_normal((n, 384)).astype(np.float32)
    idx, sim, st = _stats(ops, torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda())
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    idx_ref, sim_ref = orc.match_ip_top1(qn, bn)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_array_equal(sim.cpu().numpy(), sim_ref)
    assert st[1] > n // 4, st          # the refinement really ran for many queries
    assert st[0] <= n // 50, st        # ... and the all-pairs fallback stayed the exception


def test_many_exact_duplicates_and_long_lists_need_no_all_pairs_fallback():
    """> 64 rows tying within the fp32 margin (100 exact duplicates): the refinement cannot thin them, the whole list goes
    to the fp64 decision; ~290 candidate chunks in the window (beyond round 1's cap of 40, below the list capacity
    min(#chunks, 2048)) are refined.  Neither needs the all-pairs kernel; both give the oracle's answer (ties -> lowest
    index)."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(3)
    d, m = 384, 300 * 128
    b = rng.standard_normal((m, d)).astype(np.float32)
    base = rng.standard_normal(d).astype(np.float32)
    dup_rows = np.sort(rng.choice(m, 100, replace=False))
    b[dup_rows] = base                                            # 100 exact duplicates -> refine keeps > 64
    base2 = rng.standard_normal(d).astype(np.float32)
    rows2 = np.arange(290) * 128 + rng.integers(0, 128, 290)
    rows2 = rows2[~np.isin(rows2, dup_rows)]
    # ~290 chunks inside the fp16 window (cosines spread by ~1e-5: wider than the fp32 margin, so the refinement thins them)
    b[rows2] = base2 + 2e-2 * rng.standard_normal((len(rows2), d)).astype(np.float32)
    q = rng.standard_normal((200, d)).astype(np.float32)
    q[:20] = base + 1e-3 * rng.standard_normal((20, d)).astype(np.float32)
    q[20:40] = base2 + 1e-3 * rng.standard_normal((20, d)).astype(np.float32)
    idx, sim, st = _stats(ops, torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda())
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    idx_ref, sim_ref = orc.match_ip_top1(qn, bn)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_array_equal(sim.cpu().numpy(), sim_ref)
    assert (idx[:20].cpu().numpy() == dup_rows[0]).all()
    assert st[0] == 0, st                                         # no all-pairs fallback
    assert st[8 + 9] >= 20, st                                    # 20 queries with 257..512 candidate entries


def test_select_overflow_beyond_list_capacity_falls_back_exactly():
    """a map with more than 2048 chunks and a direction duplicated into > 2048 of them"""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(4)
    d, m = 128, 2200 * 128
    b = rng.standard_normal((m, d)).astype(np.float32)
    base = rng.standard_normal(d).astype(np.float32)
    rows = np.arange(2100) * 128 + rng.integers(0, 128, 2100)
    b[rows] = base + 1e-4 * rng.standard_normal((2100, d)).astype(np.float32)
    q = rng.standard_normal((70, d)).astype(np.float32)
    q[:6] = base + 1e-3 * rng.standard_normal((6, d)).astype(np.float32)
    idx, sim, st = _stats(ops, torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda())
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    idx_ref, sim_ref = orc.match_ip_top1(qn, bn)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_array_equal(sim.cpu().numpy(), sim_ref)
    assert st[0] == 6, st


def test_zero_rows_do_not_flood_the_record_lists():
    """points seen by no camera carry an all-zero descriptor (prepare_scenes.py:102-104): such a query scores exactly
    2.0 against every map row in the coarse pass and must not record anything (index 0, similarity 0 as the oracle gives);
    zero MAP rows must not disturb the others either."""
    from oracle import oracle as orc
    from vfmreg import ops
    rng = np.random.default_rng(8)
    n, m, d = 3000, 40000, 384
    b = rng.standard_normal((m, d)).astype(np.float32)
    b[rng.choice(m, 2000, replace=False)] = 0.0
    q = b[rng.permutation(m)[:n]] + 0.1 * rng.standard_normal((n, d)).astype(np.float32)
    zero_q = rng.choice(n, n // 2, replace=False)
    q[zero_q] = 0.0
    idx, sim, st = _stats(ops, torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda())
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    idx_ref, sim_ref = orc.match_ip_top1(qn, bn)
    np.testing.assert_array_equal(idx.cpu().numpy(), idx_ref)
    np.testing.assert_array_equal(sim.cpu().numpy(), sim_ref)
    assert (idx[zero_q].cpu().numpy() == 0).all() and (sim[zero_q].cpu().numpy() == 0).all()
    assert st[0] == 0 and st[4] < 60 * n, st      # a few dozen records per non-zero query, none for the zero ones
