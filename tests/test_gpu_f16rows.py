"""Descriptor rows stored in fp16 (round 5; BASELINE.json configs[4] "fp16 descriptor storage"): include/vfmreg.h
VFM_ROWS_F16.  An fp16 row is widened to fp32 element by element as the kernels load it -- preparation (norms, int8 / fp6 images) and
finish stage (fp32 refinement, fp64 decision) -- so a search on fp16 rows must equal, bit for bit, the search on rows.astype(float32):
against the fp32 entry points on the widened rows (every record kind) and against the oracle on the widened rows
(VoxelHashMap.cpp:469-511 on what faiss would be handed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as orc          # noqa: E402
from vfmreg import _lib, synth            # noqa: E402
from vfmreg.pipeline import RegistrationPipeline   # noqa: E402

PREPARE_MX6, PREPARE_MX6_HALF = 8, 16


def _search(q, b, gate, records, flags):
    """the gated search through the _t entry points where an operand is fp16, the fp32 ones otherwise"""
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    fq, fb = int(q.dtype == torch.float16), int(b.dtype == torch.float16)
    qb = torch.zeros(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")   # (zeroed: the buffers are compared byte for byte)
    bb = torch.zeros(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    if fq or fb:
        _lib.check(lib.vfm_match_prepare2_gated_t(b.data_ptr(), fb, m, bb.data_ptr(), q.data_ptr(), fq, n, qb.data_ptr(), d, flags, st))
    else:
        _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st))
    _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
    if fq or fb:
        _lib.check(lib.vfm_match_search_finish_gated_t(q.data_ptr(), fq, qb.data_ptr(), n, b.data_ptr(), fb, bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    else:
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    torch.cuda.synchronize()
    return idx, sim, qb, bb


@pytest.mark.parametrize("d,n,m", [(384, 3000, 20000), (256, 2500, 7001), (768, 2300, 9000), (512, 2200, 5000), (384, 2100, 130)])
def test_search_on_fp16_rows_equals_the_search_on_the_widened_rows(d, n, m):
    rng = np.random.default_rng(d + n)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    q[::3] = rng.standard_normal((len(q[::3]), d)).astype(np.float32)
    b[7] = 0.0                                   # a zero row stays a zero row
    q[5] = 0.0
    b[11] = 3.0e-5 * b[11]                       # rows of fp16 subnormals
    b16, q16 = torch.from_numpy(b).cuda().half(), torch.from_numpy(q).cuda().half()
    bw, qw = b16.float().contiguous(), q16.float().contiguous()           # the widened rows: what the kernels must see
    q32 = torch.from_numpy(q).cuda()
    # oracle on the widened rows (map fp16, scan fp16) and on (map fp16, scan fp32)
    refs = {}
    for name, qq in (("both", qw), ("map", q32)):
        qn, _ = orc.l2norm_rows(qq.cpu().numpy())
        bn, _ = orc.l2norm_rows(bw.cpu().numpy())
        refs[name] = orc.match_ip_top1(qn, bn)
    kinds = [(0, 0), (1, 0), (3, 0), (4, 0), (5, PREPARE_MX6), (7, PREPARE_MX6), (8, PREPARE_MX6 | PREPARE_MX6_HALF), (10, PREPARE_MX6)]
    for records, flags in kinds:
        for name, qh, qf in (("both", q16, qw), ("map", q32, q32)):
            i16, s16, qb16, bb16 = _search(qh, b16, gate, records, flags)
            i32, s32, qb32, bb32 = _search(qf, bw, gate, records, flags)
            assert torch.equal(i16, i32) and torch.equal(s16, s32), (records, name, int((i16 != i32).sum()))
            if records == 0:
                # the prepared operands themselves (norms, images, bounds): the same bytes as from the widened rows -- fp32 rows take
                # prep_stream_kernel for the fp6 kinds, fp16 rows prep_chunk_kernel (byte-identical forms: tests/test_gpu_mx6.py)
                assert torch.equal(bb16, bb32) and torch.equal(qb16, qb32), (records, name)
            ridx, rsim = refs[name]
            gi, gs = i16.cpu().numpy(), s16.cpu().numpy()
            solved = gi >= 0
            np.testing.assert_array_equal(gi[solved], ridx[solved], err_msg=f"{records} {name}")
            np.testing.assert_array_equal(gs[solved], rsim[solved], err_msg=f"{records} {name}")
            assert (rsim[~solved] < 0.8).all() and solved[rsim >= 0.8].all(), (records, name)


def test_fp16_rows_are_refused_where_no_kernel_widens_them():
    """the fp16-tile pass of small searches (and d = 128) reads fp32 rows: VFM_EINVAL, not a wrong answer"""
    lib = _lib.load()
    n, m, d = 64, 500, 384            # below the int8 pass' query count
    q = torch.randn(n, d, device="cuda").half()
    b = torch.randn(m, d, device="cuda").half()
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.vfm_match_prepare2_gated_t(b.data_ptr(), 1, m, bb.data_ptr(), q.data_ptr(), 1, n, qb.data_ptr(), d, 0, st)
    torch.cuda.synchronize()
    if rc == 0:    # (the gated family may run the int8 pass at every size: then the search must simply be right)
        ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
        idx = torch.empty(n, dtype=torch.int64, device="cuda")
        sim = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 0, float("-inf"), st))
        _lib.check(lib.vfm_match_search_finish_gated_t(q.data_ptr(), 1, qb.data_ptr(), n, b.data_ptr(), 1, bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), float("-inf"), 0, st))
        torch.cuda.synchronize()
        qn, _ = orc.l2norm_rows(q.float().cpu().numpy())
        bn, _ = orc.l2norm_rows(b.float().cpu().numpy())
        ridx, rsim = orc.match_ip_top1(qn, bn)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    else:
        assert b"fp16 rows" in lib.vfm_last_error()
    assert lib.vfm_match_prepare2_gated_t(b.data_ptr(), 7, m, bb.data_ptr(), q.data_ptr(), 1, n, qb.data_ptr(), d, 0, st) != 0


@pytest.mark.parametrize("coarse", ["auto", "mx6-half", "int8"])
def test_pipeline_with_an_fp16_map_equals_the_pipeline_on_the_widened_rows_and_the_oracle(coarse):
    n, m, d, iters = 3000, 20000, 384, 2000
    p = synth.make_pair_device(n, m, d, seed=23)
    b16 = p["b_desc"].half().contiguous()
    bw = b16.float().contiguous()
    outs = {}
    for name, bdesc in (("fp16 map", b16), ("widened", bw)):
        pipe = RegistrationPipeline(n, m, d, n_iter=iters, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=coarse)
        out = None
        for _ in range(4):
            out = pipe.register(p["q_desc"], p["q_xyz"], bdesc, p["b_xyz"])
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
        outs[name] = {k: out[k].clone() for k in ("T", "count", "corres", "mask", "best_hyp", "idx", "sim")}
        del pipe
    a, w = outs["fp16 map"], outs["widened"]
    c = int(w["count"].item())
    assert c > 1000 and int(a["count"].item()) == c
    for k in ("T", "best_hyp"):
        assert torch.equal(a[k], w[k]), k
    assert torch.equal(a["corres"][:c], w["corres"][:c]) and torch.equal(a["mask"][:c], w["mask"][:c])
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(bw.cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    keep = ~(rsim.astype(np.float64) < 0.8)
    corres = np.stack([np.nonzero(keep)[0], ridx[keep]], 1).astype(np.int32)
    ref = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, iters, seed=42)
    np.testing.assert_array_equal(a["corres"][:c].cpu().numpy(), corres)
    np.testing.assert_array_equal(a["T"].cpu().numpy(), ref.transformation)
    np.testing.assert_array_equal(a["mask"][:c].cpu().numpy(), ref.inlier_mask)
    with pytest.raises(ValueError):
        RegistrationPipeline(n, m, d, n_iter=iters, coarse="fp16").register(p["q_desc"], p["q_xyz"], b16, p["b_xyz"])


def test_auto_policy_never_hands_fp16_rows_to_the_fp32_kernels():
    """`auto` may leave the int8 passes for the fp16 one from the searches' feedback (above TOP2_LIMIT rescans per
    query: duplicate-rich maps).  That decision is taken inside register() -- after the dtype check of round 5 -- and the fp16 pass
    reads float32 rows: float16 storage then went to kernels that read twice as far.  The policy is now read after it is updated and
    fp16 rows stay on the int8 pass with top-2 records; the registration is that of the widened rows."""
    n, m, d, iters = 3000, 20000, 384, 2000
    p = synth.make_pair_device(n, m, d, seed=29)
    # a duplicate-rich map: every row eight times (the rescan count per query goes up with it)
    b = p["b_desc"][: m // 8].repeat(8, 1).contiguous()
    bx = p["b_xyz"][: m // 8].repeat(8, 1).contiguous()
    b16 = b.half().contiguous()
    bw = b16.float().contiguous()
    outs = {}
    for name, bdesc in (("fp16 map", b16), ("widened", bw)):
        pipe = RegistrationPipeline(n, m, d, n_iter=iters, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="auto")
        pipe.HALF_LIMIT = pipe.RESCAN_LIMIT = pipe.MX6_UP = -1.0     # every feedback says "too many": auto walks the whole ladder
        pipe.TOP2_LIMIT = -1.0
        seen_fp16_pass = False
        out = None
        for _ in range(8):
            out = pipe.register(p["q_desc"], p["q_xyz"], bdesc, bx)
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
            seen_fp16_pass |= not pipe.use_i8
        assert seen_fp16_pass, "the policy never asked for the fp16 pass: the test does not exercise the hand-over"
        outs[name] = {k: out[k].clone() for k in ("T", "count", "corres", "idx", "sim")}
        del pipe
    a, w = outs["fp16 map"], outs["widened"]
    c = int(w["count"].item())
    assert int(a["count"].item()) == c
    assert torch.equal(a["T"], w["T"]) and torch.equal(a["corres"][:c], w["corres"][:c])
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(bw.cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    gi = a["idx"].cpu().numpy()
    solved = gi >= 0
    np.testing.assert_array_equal(gi[solved], ridx[solved])
    assert (rsim[~solved] < 0.8).all()
