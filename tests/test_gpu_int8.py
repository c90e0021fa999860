"""The int8 coarse pass (d = 256 ... 768, the gated family of entry points): its quantisation bound checked pair by pair against fp64
scores, oracle-identical answers on inputs that stress the quantisation, and the contract of the similarity gate."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as orc  # noqa: E402
from vfmreg import _lib, ops, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402


def _i8_rows(P):
    lib = _lib.load()
    q8 = np.empty((P.rows, P.d), np.int8)
    step, err, gerr = (np.empty(P.rows, np.float32) for _ in range(3))
    _lib.check(lib.vfm_debug_i8_rows(P.buf.data_ptr(), P.rows, P.d, q8.ctypes.data, step.ctypes.data, err.ctypes.data, gerr.ctypes.data))
    return q8, step, err, gerr


def _search_gated(q, b, gate, records):
    """prepare -> coarse -> finish of the gated family with an explicit record kind (0 = best score, 1 = packed top-2)."""
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    Q, B = ops.PreparedRows(q), ops.PreparedRows(b)
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vfm_match_search_coarse_gated_r(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, st))
    _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), Q.buf.data_ptr(), n, b.data_ptr(), B.buf.data_ptr(), m, d, idx.data_ptr(),
                                                   sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    torch.cuda.synchronize()
    return idx, sim


def _heavy_tailed(rng, rows, d):
    x = rng.standard_normal((rows, d)).astype(np.float32)
    x[rng.random((rows, d)) < 0.002] *= 12.0            # outlier elements
    x[::97] = 0.0
    x[::97, rng.integers(0, d, len(x[::97]))] = 1.0     # one-hot rows
    x[5::131] *= 1e-18                                  # tiny norms
    x[7::89] = x[6::89][: len(x[7::89])]                # exact duplicates
    return x


@pytest.mark.parametrize("d", [256, 384, 512, 640, 768])
def test_quantisation_bound_holds_for_every_pair(d):
    """| v_a . v_b - s_a s_b (q_a . q_b) | <= (1 + 2^-13 + E_a) E_b + (1 + 2^-13) E_a for every pair, with the kernel's own
    steps, integers and measured residual norms (csrc/match_prep.hip, prep_chunk_kernel): Gaussian and heavy-tailed rows."""
    rng = np.random.default_rng(d)
    n, m = 700, 3000
    gens = ((lambda r: rng.standard_normal((r, d)).astype(np.float32), 0.03), (lambda r: _heavy_tailed(rng, r, d), 0.2))
    for gen, typical in gens:
        q, b = gen(n), gen(m)
        Q, B = ops.PreparedRows(torch.from_numpy(q).cuda()), ops.PreparedRows(torch.from_numpy(b).cuda())
        q8, sq, eq, _ = _i8_rows(Q)
        b8, sb, eb, gb = _i8_rows(B)
        vq, _ = orc.l2norm_rows(q)
        vb, _ = orc.l2norm_rows(b)
        # the measured E really bounds the residual of the oracle's normalised rows
        for v8, s, e, v in ((q8, sq, eq, vq), (b8, sb, eb, vb)):
            res = np.linalg.norm(v.astype(np.float64) - s[:, None].astype(np.float64) * v8.astype(np.float64), axis=1)
            assert (e.astype(np.float64) >= res).all()
            assert (np.abs(v8.astype(np.int32)) <= 127).all()
        assert (gb >= eb).all()
        t = vq.astype(np.float64) @ vb.astype(np.float64).T
        S = q8.astype(np.float64) @ b8.astype(np.float64).T
        dev = np.abs(t - sq[:, None].astype(np.float64) * sb[None, :].astype(np.float64) * S)
        bound = (1 + 2.0 ** -13 + eq[:, None].astype(np.float64)) * gb[None, :] + (1 + 2.0 ** -13) * eq[:, None].astype(np.float64)
        assert (dev <= bound).all()
        # ... and is not vacuous: ~2e-2 on Gaussian unit rows; a one-hot row coarsens the step of its whole 128-row group
        assert np.median(bound) < typical


@pytest.mark.parametrize("d,n,m", [(384, 1500, 9000), (256, 2050, 5003), (384, 777, 130), (512, 900, 4100), (640, 1030, 3000),
                                   (768, 1300, 6000), (384, 1, 200), (256, 37, 1000), (384, 300, 20011), (384, 2049, 129)])
def test_int8_search_equals_the_oracle_on_stress_inputs(d, n, m):
    rng = np.random.default_rng(n + m)
    cases = {
        "heavy": (_heavy_tailed(rng, n, d), _heavy_tailed(rng, m, d)),
        "anticorrelated": (None, None),
        "duplicates": (None, None),
    }
    base = rng.standard_normal((1, d)).astype(np.float32)
    bq = -base + 0.3 * rng.standard_normal((n, d)).astype(np.float32)            # every similarity negative
    bb = base + 0.3 * rng.standard_normal((m, d)).astype(np.float32)
    cases["anticorrelated"] = (bq, bb)
    few = rng.standard_normal((16, d)).astype(np.float32)
    cases["duplicates"] = (few[rng.integers(0, 16, n)] + 0.0, few[rng.integers(0, 16, m)] + 0.0)  # m rows, 16 distinct
    for name, (q, b) in cases.items():
        qn, _ = orc.l2norm_rows(q)
        bn, _ = orc.l2norm_rows(b)
        ridx, rsim = orc.match_ip_top1(qn, bn)
        qd, bd = torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda()
        # the gated family with gate = -inf: the int8 pass, every query resolved -- one-shot call, then both record kinds
        results = [ops.match_ip_top1(qd, bd, ops.FAST, gate=float("-inf")), _search_gated(qd, bd, float("-inf"), 0),
                   _search_gated(qd, bd, float("-inf"), 1)]
        for kind, (idx, sim) in enumerate(results):
            np.testing.assert_array_equal(idx.cpu().numpy(), ridx, err_msg=f"{name} / call {kind}")
            np.testing.assert_array_equal(sim.cpu().numpy(), rsim, err_msg=f"{name} / call {kind}")


def test_gate_leaves_only_provably_rejected_queries_unresolved():
    n, m, d = 4000, 30000, 384
    p = synth.make_pair_device(n, m, d, seed=5)
    outs = {}
    for gate in (True, False):
        pipe = RegistrationPipeline(n, m, d, n_iter=20000, gate=gate)
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        torch.cuda.synchronize()
        outs[gate] = {k: out[k].clone() for k in ("T", "idx", "sim", "count", "corres", "mask", "best_hyp")}
    g, f = outs[True], outs[False]
    k = int(f["count"].item())
    assert k == int(g["count"].item()) > 1000
    for key in ("T", "best_hyp"):
        assert torch.equal(g[key], f[key]), key
    for key in ("corres", "mask"):
        assert torch.equal(g[key][:k], f[key][:k]), key
    unresolved = g["idx"] < 0
    assert int(unresolved.sum()) > n // 4                      # the planted outliers (50 %) mostly end here
    assert torch.equal(g["idx"][~unresolved], f["idx"][~unresolved]) and torch.equal(g["sim"][~unresolved], f["sim"][~unresolved])
    assert bool((f["sim"][unresolved] < 0.8).all()) and bool((g["sim"][unresolved] == -2.0).all())
    # the ungated one-shot call (fp16 pass) and the gated one with a real gate agree wherever the latter resolves
    i0, s0 = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST)
    i1, s1 = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST, gate=0.5)
    ok = i1 >= 0
    assert torch.equal(i0[ok], i1[ok]) and torch.equal(s0[ok], s1[ok]) and bool((s0[~ok] < 0.5).all()) and int((~ok).sum()) > n // 4
    assert torch.equal(i0, f["idx"]) and torch.equal(s0, f["sim"])
    # and the fully resolved run is the oracle's
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    np.testing.assert_array_equal(f["idx"].cpu().numpy(), ridx)
    np.testing.assert_array_equal(f["sim"].cpu().numpy(), rsim)


def test_pipeline_changes_records_on_a_duplicate_rich_map_and_results_do_not_change():
    """auto mode: the first gated search reports how many candidate chunks it had to rescan; on a map where every point has
    about eighty near-copies the pipeline -- with the switch point set to 20 rescanned chunks per query here; the fitted value sits
    beyond what this small map can produce -- moves to the packed top-2 records (a chunk with one row inside the bounds then costs
    one fp32 row instead of a 48 KB rescan).  Poses and correspondences are those of every fixed mode."""
    n, m, d = 2000, 40000, 384
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    p = synth.make_pair_device(n, m, d, seed=9)
    phys = torch.randn((500, d), generator=g, device="cuda")       # every physical point ~80 times in the map
    owner = torch.randint(0, 500, (m,), generator=g, device="cuda")
    b = phys[owner] + 0.01 * torch.randn((m, d), generator=g, device="cuda") / d ** 0.5
    q = b[p["match"].clamp(min=0)] + 0.02 * torch.randn((n, d), generator=g, device="cuda") / d ** 0.5
    q = torch.where((p["match"] < 0)[:, None], torch.randn((n, d), generator=g, device="cuda"), q)
    b, q = b.contiguous(), q.contiguous()
    outs = {}
    for coarse in ("auto", "int8", "int8-top2", "fp16"):
        pipe = RegistrationPipeline(n, m, d, n_iter=5000, overlap_ransac=True, coarse=coarse)
        pipe.RESCAN_LIMIT = 20.0
        first_rescans = None
        for _ in range(5):
            out = pipe.register(q, p["q_xyz"], b, p["b_xyz"])
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
            if first_rescans is None:
                first_rescans = pipe.last_rescans
        k = int(out["count"].item())
        outs[coarse] = (out["T"].clone(), out["corres"][:k].clone(), pipe.use_i8, pipe.top2, first_rescans, pipe.half)
        del pipe
    # the int8 pass stays; best-score records do not: the feedback moves on to top-2 records, or -- where the probe of the
    # half-width pass finds few enough survivors (the ~80 copies of a matched point) -- to the half-width pass
    assert outs["auto"][2] is True and (outs["auto"][3] is True or outs["auto"][5] is True)
    assert outs["auto"][4] > 20.0 * n                                      # what the first (best-score) search reported
    assert outs["int8"][2:4] == (True, False) and outs["int8-top2"][2:4] == (True, True) and outs["fp16"][2] is False
    for coarse in ("int8", "int8-top2", "fp16"):
        assert torch.equal(outs["auto"][0], outs[coarse][0]) and torch.equal(outs["auto"][1], outs[coarse][1]), coarse
    assert outs["auto"][1].shape[0] > 500


def test_quantisation_bound_at_c2_scale():
    """The same pair-by-pair check at the benchmark's size: every 10th query of C2 against all 200 000 map rows, fp64
    scores of the oracle's normalised rows computed on the device (the checker, not the product)."""
    n, m, d = 20000, 200000, 384
    p = synth.make_pair_device(n, m, d, seed=42)
    Q, B = ops.PreparedRows(p["q_desc"]), ops.PreparedRows(p["b_desc"])
    q8, sq, eq, _ = _i8_rows(Q)
    b8, sb, eb, gb = _i8_rows(B)
    assert float(np.median(eb)) < 0.012 and float(gb.max()) < 0.02           # Gaussian unit rows: E ~ 0.0098
    dev = torch.device("cuda")
    rows = torch.arange(0, n, 10, device=dev)
    # the oracle's normalisation (fp32 sum of squares in its order is what prep_chunk_kernel reproduces; here the rows are
    # re-normalised in fp64 from the kernel's own 1/|row|, which test_gpu_parity pins to the oracle bit for bit)
    idx, sim = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST, gate=float("-inf"))
    vq = (p["q_desc"][rows] * (1.0 / p["q_desc"][rows].double().norm(dim=1, keepdim=True)).float()).double()
    worst_slack, worst_dev = float("inf"), 0.0
    tq8 = torch.from_numpy(q8).to(dev)[rows].double()
    tsq, teq = torch.from_numpy(sq).to(dev)[rows].double(), torch.from_numpy(eq).to(dev)[rows].double()
    best = torch.full((len(rows),), -2.0, dtype=torch.float64, device=dev)
    arg = torch.zeros(len(rows), dtype=torch.int64, device=dev)
    for c0 in range(0, m, 20000):
        bb = p["b_desc"][c0:c0 + 20000]
        vb = (bb * (1.0 / bb.double().norm(dim=1, keepdim=True)).float()).double()
        t = vq @ vb.T
        S = tq8 @ torch.from_numpy(b8[c0:c0 + 20000]).to(dev).double().T
        tsb = torch.from_numpy(sb[c0:c0 + 20000]).to(dev).double()
        tgb = torch.from_numpy(gb[c0:c0 + 20000]).to(dev).double()
        devn = (t - tsq[:, None] * tsb[None, :] * S).abs()
        bound = (1 + 2.0 ** -13 + teq[:, None]) * tgb[None, :] + (1 + 2.0 ** -13) * teq[:, None]
        # the re-normalisation above differs from the kernel's fp32 one by < 2e-7 per score: far inside the bound's slack
        worst_slack = min(worst_slack, float((bound - devn).min()))
        worst_dev = max(worst_dev, float(devn.max()))
        cb, ca = t.max(dim=1)
        upd = cb > best
        best = torch.where(upd, cb, best)
        arg = torch.where(upd, ca + c0, arg)
    assert worst_slack > 0.0, (worst_slack, worst_dev)
    assert worst_dev < 0.01                                                    # the actual error is a fraction of the bound
    # and the search's answers on those rows are the arg-max of the fp64 scores (up to exact ties, which the data has none of)
    assert torch.equal(idx[rows], arg)


def _gate_contract(idx, sim, ridx, rsim, gate):
    idx, sim = idx.cpu().numpy(), sim.cpu().numpy()
    solved = idx >= 0
    np.testing.assert_array_equal(idx[solved], ridx[solved])
    np.testing.assert_array_equal(sim[solved], rsim[solved])
    assert (rsim[~solved] < gate).all() and (sim[~solved] == -2.0).all()
    return solved


@pytest.mark.parametrize("d", [384, 768])
def test_chunk_major_rescan_full_bins_and_padded_chunks(d):
    """Best-score records with many queries per map chunk take the chunk-major rescan (match_rescan_chunk_kernel): bins that
    overflow (the rest of the chunk's queries stay query-major), matches inside the partly padded last chunk (not covered by
    the running lower bound: the "provably below the gate" shortcut of match_select_best_kernel must not drop them), gate
    off / on, and the same answers with the chunk-major path switched off (variant 21) and from the general selection
    kernel (variant 20)."""
    lib = _lib.load()
    rng = np.random.default_rng(d)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    # (a) 8 chunks, 6000 matched queries: 750 per chunk against a bin of 512
    m, n = 1000, 6000
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.25 * rng.standard_normal((n, d)).astype(np.float32)
    q[::7] = rng.standard_normal((len(q[::7]), d)).astype(np.float32)       # unmatched queries
    # (b) 20 full chunks + 40 rows: half of the queries match rows of the padded chunk
    m2, n2 = 128 * 20 + 40, 600
    b2 = rng.standard_normal((m2, d)).astype(np.float32)
    q2 = rng.standard_normal((n2, d)).astype(np.float32)
    q2[::2] = b2[rng.integers(128 * 20, m2, len(q2[::2]))] + 0.2 * rng.standard_normal((len(q2[::2]), d)).astype(np.float32)
    for name, (qq, bb) in {"full bins": (q, b), "padded chunk": (q2, b2)}.items():
        qn, _ = orc.l2norm_rows(qq)
        bn, _ = orc.l2norm_rows(bb)
        ridx, rsim = orc.match_ip_top1(qn, bn)
        qd, bd = torch.from_numpy(qq).cuda(), torch.from_numpy(bb).cuda()
        ref = {}
        for variant in (0, 21, 20):
            lib.vfm_debug_set_coarse_variant(variant)
            try:
                i0, s0 = _search_gated(qd, bd, float("-inf"), 0)
                i1, s1 = _search_gated(qd, bd, gate, 0)
            finally:
                lib.vfm_debug_set_coarse_variant(0)
            np.testing.assert_array_equal(i0.cpu().numpy(), ridx, err_msg=f"{name} / variant {variant}")
            np.testing.assert_array_equal(s0.cpu().numpy(), rsim, err_msg=f"{name} / variant {variant}")
            solved = _gate_contract(i1, s1, ridx, rsim, gate)
            assert solved[rsim >= 0.8].all() and solved.sum() > len(solved) // 3 and (~solved).sum() > 0
            ref.setdefault("solved", solved)
            np.testing.assert_array_equal(solved, ref["solved"], err_msg=f"{name} / variant {variant}")


def _search_gated_split(q, b, gate, records, want_stats=False):
    """The gated family exactly as the pipeline calls it: vfm_match_prepare2_gated (the only prepare that writes the half-width
    image) -> coarse_gated_r -> finish_gated_r.  ``want_stats``: also the search's 64 counters (fb_count)."""
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vfm_match_prepare2_gated(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, st))
    _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
    _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                   sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    torch.cuda.synchronize()
    if want_stats:
        st64 = (C.c_int32 * 64)()
        _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, C.cast(st64, C.c_void_p)))
        return idx, sim, list(st64)
    return idx, sim


@pytest.mark.parametrize("d,n,m", [(384, 2500, 9000), (256, 3000, 5003), (384, 2100, 130), (512, 900, 4100), (768, 1300, 6000),
                                   (384, 300, 20011), (640, 1030, 3000)])
def test_half_width_pass_keeps_the_gate_contract(d, n, m):
    """VFM_RECORDS_HALF: int8 scores over the first d / 2 columns + |rest of the query| * max |rest of a row of the chunk| decide
    which chunks can hold a row at the gate.  The contract: every query resolved has the oracle's index and similarity, every
    query left unresolved has an oracle similarity below the gate, and every match at or above min_cosine is resolved -- on
    planted matches, on descriptors whose energy sits entirely in one half of the columns (the bound is then exact on one side
    and vacuous on the other), on heavy-tailed rows, on rows that are all alike (everything survives) and on exact duplicates.
    (384, 300, .) and (640, ., .) run the one-set kernel on half-width tiles.)"""
    rng = np.random.default_rng(d + n)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    cases = {}
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    q[::3] = rng.standard_normal((len(q[::3]), d)).astype(np.float32)
    cases["planted"] = (q, b)
    b2, q2 = b.copy(), q.copy()
    b2[: m // 2, : d // 2] *= 1e-3      # half of the map lives in the second half of the columns ...
    b2[m // 2:, d // 2:] *= 1e-3        # ... the other half in the first
    q2[::2, : d // 2] *= 1e-3
    q2[1::2, d // 2:] *= 1e-3
    cases["energy in one half"] = (q2, b2)
    cases["heavy-tailed"] = (_heavy_tailed(rng, n, d), _heavy_tailed(rng, m, d))
    base = rng.standard_normal((1, d)).astype(np.float32)
    cases["all alike"] = (base + 0.2 * rng.standard_normal((n, d)).astype(np.float32), base + 0.2 * rng.standard_normal((m, d)).astype(np.float32))
    few = rng.standard_normal((16, d)).astype(np.float32)
    cases["duplicates"] = (few[rng.integers(0, 16, n)] + 0.0, few[rng.integers(0, 16, m)] + 0.0)
    for name, (qq, bb) in cases.items():
        qn, _ = orc.l2norm_rows(qq)
        bn, _ = orc.l2norm_rows(bb)
        ridx, rsim = orc.match_ip_top1(qn, bn)
        qd, bd = torch.from_numpy(qq).cuda(), torch.from_numpy(bb).cuda()
        i0, s0 = _search_gated_split(qd, bd, gate, 0)
        keep0 = (s0 >= 0.8).cpu().numpy()
        for records in (3, 4):   # 4 = the selection fused into the coarse kernel (where that kernel exists; else as 3)
            idx, sim = _search_gated_split(qd, bd, gate, records)
            solved = _gate_contract(idx, sim, ridx, rsim, gate)
            assert solved[rsim >= 0.8].all(), (name, records)
            # ... and the correspondences a caller keeps (similarity >= min_cosine) are those of best-score records
            keep3 = (sim >= 0.8).cpu().numpy()
            np.testing.assert_array_equal(keep3, keep0, err_msg=f"{name} / {records}")
            np.testing.assert_array_equal(idx.cpu().numpy()[keep3], i0.cpu().numpy()[keep0], err_msg=f"{name} / {records}")


def test_half_width_pass_needs_a_gate_and_the_pipeline_leaves_it_on_descriptors_that_are_all_alike():
    lib = _lib.load()
    n, m, d = 2200, 20000, 384
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    q = torch.randn((n, d), generator=g, device="cuda")
    b = torch.randn((m, d), generator=g, device="cuda")
    with pytest.raises(RuntimeError, match="finite gate"):
        _search_gated_split(q, b, float("-inf"), 3)
    # D.2-like pair: the pipeline stays on the half-width pass; a map whose rows are all alike (cosine ~0.96 to everything):
    # every chunk survives the half-width bound, the feedback moves the pipeline on, results unchanged
    p = synth.make_pair_device(n, m, d, seed=21)
    pipe = RegistrationPipeline(n, m, d, n_iter=4000, overlap_ransac=True)
    for _ in range(4):
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
    assert pipe.half and pipe.use_i8 and int(out["count"].item()) > 500
    base = torch.randn((1, d), generator=g, device="cuda")
    b2 = (base + 0.2 * torch.randn((m, d), generator=g, device="cuda")).contiguous()
    q2 = b2[p["match"].clamp(min=0)] + 0.05 * torch.randn((n, d), generator=g, device="cuda")
    q2 = torch.where((p["match"] < 0)[:, None], base + 0.2 * torch.randn((n, d), generator=g, device="cuda"), q2).contiguous()
    outs = {}
    for coarse in ("auto", "int8-half", "int8", "fp16"):
        pipe = RegistrationPipeline(n, m, d, n_iter=4000, overlap_ransac=True, coarse=coarse)
        for _ in range(4):
            out = pipe.register(q2, p["q_xyz"], b2, p["b_xyz"])
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
        k = int(out["count"].item())
        outs[coarse] = (out["T"].clone(), out["corres"][:k].clone(), pipe.half)
    assert outs["auto"][2] is False and outs["int8-half"][2] is True     # the feedback left the half-width pass
    for coarse in ("int8-half", "int8", "fp16"):
        assert torch.equal(outs["auto"][0], outs[coarse][0]) and torch.equal(outs["auto"][1], outs[coarse][1]), coarse
    assert outs["auto"][1].shape[0] > 500


def test_half_width_probe_counts_the_survivors_and_prepare_schedules_write_the_same_bytes():
    """vfm_match_search_probe_half reports exactly the load figure the half-width search itself reports afterwards
    (vfm_match_search_rescans_async); the prepare kernel's launch shapes (vfm_match_prepare2_gated_p) and the plain gated prepare
    write identical bytes; unknown record kinds / schedules are refused."""
    lib = _lib.load()
    n, m, d = 3000, 30000, 384
    p = synth.make_pair_device(n, m, d, seed=3)
    q, b = p["q_desc"], p["b_desc"]
    st = torch.cuda.current_stream().cuda_stream
    bufs = {}
    for schedule in (None, 0, 1, 2):
        qb = torch.zeros(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
        bb = torch.zeros(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
        if schedule is None:
            _lib.check(lib.vfm_match_prepare2_gated(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, st))
        else:
            _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, schedule, st))
        torch.cuda.synchronize()
        bufs[schedule] = (qb, bb)
    for schedule in (0, 1, 2):
        assert torch.equal(bufs[schedule][0], bufs[None][0]) and torch.equal(bufs[schedule][1], bufs[None][1]), schedule
    assert lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bufs[0][1].data_ptr(), q.data_ptr(), n, bufs[0][0].data_ptr(), d, 7, st) != 0
    qb, bb = bufs[None]
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    probe = torch.zeros(1, dtype=torch.int32).pin_memory()
    _lib.check(lib.vfm_match_search_probe_half(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), gate, probe.data_ptr(), st))
    torch.cuda.synchronize()
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    counts = {}
    for records in (3, 4):
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
        slot = torch.zeros(1, dtype=torch.int32).pin_memory()
        _lib.check(lib.vfm_match_search_rescans_async(ws.data_ptr(), n, m, slot.data_ptr(), st))
        torch.cuda.synchronize()
        counts[records] = int(slot.item())
    matched = int((p["match"] >= 0).sum())
    assert int(probe.item()) == counts[3] == counts[4] and matched <= counts[3] <= matched + n // 20
    assert int((sim >= 0.8).sum()) == matched
    assert lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 11, gate, st) != 0   # (10 = VFM_RECORDS_MX6_FUSED since round 5)
    with pytest.raises(RuntimeError, match="finite gate"):
        _lib.check(lib.vfm_match_search_probe_half(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), float("-inf"),
                                                   probe.data_ptr(), st))


@pytest.mark.parametrize("d,n,m", [(384, 3000, 20011), (256, 2600, 9000), (768, 1300, 12000), (384, 300, 20011)])
def test_half_width_guard_falls_through_to_the_gate_pass(d, n, m):
    """The device-side guard of the half-width pass (csrc/match_finish.hip: half_guard_kernel, match_gatepass_kernel): on
    descriptors that are all alike every (query, chunk) pair survives the half-width bound; the search must not rescan them
    (31 million 128-row rescans at C2 size: 171 ms in round 2) but fall through, inside the same _finish call, to one full-width
    pass with the gate as hit test -- flag raised, same gate contract, same kept matches as best-score records.  On planted
    matches the flag stays down."""
    rng = np.random.default_rng(d + n + m)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    base = rng.standard_normal((1, d)).astype(np.float32)
    b = base + 0.25 * rng.standard_normal((m, d)).astype(np.float32)
    pick = rng.integers(0, m, n)
    q = b[pick] + 0.05 * rng.standard_normal((n, d)).astype(np.float32)
    q[::4] = base + 0.25 * rng.standard_normal((len(q[::4]), d)).astype(np.float32)
    q[5] = 0.0                                        # a zero row (no camera saw the point): index 0, similarity 0
    b[7] = 0.0
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    ridx, rsim = orc.match_ip_top1(qn, bn)
    qd, bd = torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda()
    i0, s0 = _search_gated_split(qd, bd, gate, 0)
    keep0 = (s0 >= 0.8).cpu().numpy()
    assert keep0.sum() > n // 3
    for records in (3, 4):
        idx, sim, st = _search_gated_split(qd, bd, gate, records, want_stats=True)
        assert st[7] == 1, (records, st[:8])          # the guard flag: the search fell through
        assert st[5] > 48 * n, (records, st[5])       # the load figure the host policy reads still says "everything survives"
        solved = _gate_contract(idx, sim, ridx, rsim, gate)
        assert solved[rsim >= 0.8].all()
        keep = (sim >= 0.8).cpu().numpy()
        np.testing.assert_array_equal(keep, keep0)
        np.testing.assert_array_equal(idx.cpu().numpy()[keep], i0.cpu().numpy()[keep0])
    # planted matches on isotropic descriptors: nothing but the matches survives, the flag stays down
    p = synth.make_pair(n, m, d, seed=9)
    for records in (3, 4):
        _, sim, st = _search_gated_split(torch.from_numpy(p["q_desc"]).cuda(), torch.from_numpy(p["b_desc"]).cuda(), gate, records, want_stats=True)
        assert st[7] == 0 and st[5] < 2 * n, (records, st[:8])
        assert int((sim >= 0.8).sum()) == int((p["match"] >= 0).sum())


def test_forced_half_width_registration_on_alike_descriptors_is_bounded_at_c2_size():
    """C2 size, descriptors that are all alike, the pipeline PINNED to the half-width pass (the state a pipeline is in when the
    data change under it): round 2 measured 171-195 ms per registration there; with the device-side guard the search costs one
    full-width pass.  Same correspondences and pose as the full-width int8 pass."""
    n, m, d = 20000, 200000, 384
    p = synth.make_lifted_pair_device(n, m, d, seed=43, clouds=10, view_noise=0.1, common=1.0)
    outs = {}
    for coarse in ("int8-top2", "int8-half"):
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, coarse=coarse)
        ts = []
        for r in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        k = int(out["count"].item())
        outs[coarse] = (out["T"].clone(), out["corres"][:k].clone(), min(ts[1:]))
    print("ms per registration:", {c: round(v[2], 2) for c, v in outs.items()})
    assert torch.equal(outs["int8-half"][0], outs["int8-top2"][0]) and torch.equal(outs["int8-half"][1], outs["int8-top2"][1])
    assert outs["int8-half"][1].shape[0] > 5000
    assert outs["int8-half"][2] < 12.0, outs["int8-half"][2]     # milliseconds (round 2: 171)
