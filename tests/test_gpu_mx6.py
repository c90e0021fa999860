"""The fp6 coarse pass (VFM_RECORDS_MX6; d = 256 / 384, more than 2048 queries): the microscaled e2m3 image of
prep_chunk_kernel against its arithmetic definition, the measured residual norms, the bound they give pair by pair, and
oracle-identical answers / the gate contract of the searches that run on it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as orc  # noqa: E402
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

from .test_gpu_int8 import _gate_contract, _heavy_tailed  # noqa: E402

PREPARE_MX6 = 8
RECORDS_MX6 = 5
RECORDS_MX6_TOP2 = 6
RECORDS_MX6_PILOT = 9
RECORDS_MX6_HALF = 7
RECORDS_MX6_HALF_FUSED = 8
RECORDS_MX6_FUSED = 10     # round 5: the full-width pass with the gate test in its epilogue (no records)
PREPARE_MX6_HALF = 16
HALF_KINDS = (RECORDS_MX6_HALF, RECORDS_MX6_HALF_FUSED)
E2M3 = np.array(sorted({(mm / 8 if e == 0 else (1 + mm / 8) * 2 ** (e - 1)) for e in range(4) for mm in range(8)}))


def mx6_image(v):
    """The definition of the fp6 image (csrc/match_prep.hip): the fp32-normalised rows rounded to fp16; per 32 columns the scale
    2^e with max / 2^e <= 7.75 -- for max = (1 + f) 2^x: e = x - 2 if 1 + f <= 1.9375 else x - 1, never above 0; elements rounded
    to the nearest e2m3 value (ties to even on the binade's grid, saturating at 7.5).  float64 on exactly representable values."""
    n, d = v.shape
    b = v.astype(np.float16).reshape(n, d // 32, 32).astype(np.float64)
    amax = np.abs(b).max(-1, keepdims=True)
    x = np.floor(np.log2(np.where(amax > 0, amax, 1.0)))
    x = np.maximum(x, -14.0)                                   # fp16 denormals: exponent field 0 reads as 2^-15 below
    frac = np.where(amax >= 2.0 ** -14, amax / 2.0 ** x, 0.0)
    e = np.where(amax >= 2.0 ** -14, np.where(frac <= 1.9375, x - 2, x - 1), -17.0)
    e = np.minimum(e, 0)
    s = 2.0 ** e
    a = np.abs(b) / s
    sh = np.where(a < 2, 3, np.where(a < 4, 2, 1))
    k = np.minimum(np.rint(a * 2.0 ** sh), np.where(sh == 1, 15, 16))
    return (np.sign(b) * k / 2.0 ** sh * s).reshape(n, d)


def _prepare(b, q, flags=PREPARE_MX6):
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags,
                                              torch.cuda.current_stream().cuda_stream))
    return qb, bb


def _mx6_rows(buf, rows, d):
    lib = _lib.load()
    v6 = np.empty((rows, d), np.float32)
    err, gerr = np.empty(rows, np.float32), np.empty(rows, np.float32)
    _lib.check(lib.vfm_debug_mx6_rows(buf.data_ptr(), rows, d, v6.ctypes.data, err.ctypes.data, gerr.ctypes.data))
    return v6, err, gerr


def _search(q, b, gate, records, flags=PREPARE_MX6):
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    qb, bb = _prepare(b, q, flags)
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
    _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                   sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    torch.cuda.synchronize()
    return idx, sim


@pytest.mark.parametrize("d,n,m", [(384, 2500, 9000), (256, 3000, 777), (768, 2300, 1500), (512, 700, 2600)])
def test_mx6_image_is_the_e2m3_quantisation_of_the_fp16_rows_and_its_residuals_are_measured(d, n, m):
    """Every element of the fp6 image equals the definition above on the oracle's normalised rows (bit-identical to the
    library's); E of a row is at least the true |v - image|_2 (fp64) and at most that + the declared roundings; the group value is the
    maximum over its 128 rows; heavy-tailed rows, one-hot rows, tiny norms, zero rows included."""
    rng = np.random.default_rng(d + m)
    b = _heavy_tailed(rng, m, d)
    q = rng.standard_normal((n, d)).astype(np.float32)
    qb, bb = _prepare(torch.from_numpy(b).cuda(), torch.from_numpy(q).cuda())
    for x, buf, rows in ((b, bb, m), (q, qb, n)):
        xn, _ = orc.l2norm_rows(x)
        v6, err, gerr = _mx6_rows(buf, rows, d)
        np.testing.assert_array_equal(v6.astype(np.float64), mx6_image(xn))
        true = np.linalg.norm(xn.astype(np.float64) - v6.astype(np.float64), axis=1)
        assert (err >= true).all()
        assert (err <= true * 1.0003 + 1.1e-3).all()   # the fp16 rounding is bounded (4.9e-4), not measured, and sits inside `true` too
        gmax = np.array([err[g:g + 128].max() for g in range(0, rows, 128)])
        np.testing.assert_array_equal(gerr[::128], gmax)


def _mx6_half_err(buf, rows, d):
    lib = _lib.load()
    errh, gerrh = np.empty(rows, np.float32), np.empty(rows, np.float32)
    _lib.check(lib.vfm_debug_mx6_half_err(buf.data_ptr(), rows, d, errh.ctypes.data, gerrh.ctypes.data))
    return errh, gerrh


@pytest.mark.parametrize("d,n,m", [(384, 2500, 9000), (256, 3000, 777), (768, 2300, 1500), (512, 700, 2600)])
def test_half_width_fp6_image_and_its_bounds(d, n, m):
    """VFM_PREPARE_MX6_HALF (round 4): only the first d / 2 columns are converted -- those columns of the image equal the definition,
    the full-width E is infinite (a full-width fp6 search on such an operand prunes nothing), and E over the first d / 2 columns
    -- what the half-width kinds bound with, also written by a full VFM_PREPARE_MX6 -- is at least the true residual of those
    columns, the same bits in both forms; the bound holds pair by pair on the half dot products."""
    rng = np.random.default_rng(d + m + 1)
    b = _heavy_tailed(rng, m, d)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    bd, qd = torch.from_numpy(b).cuda(), torch.from_numpy(q).cuda()
    qb_full, bb_full = _prepare(bd, qd, PREPARE_MX6)
    qb, bb = _prepare(bd, qd, PREPARE_MX6 | PREPARE_MX6_HALF)
    h = d // 2
    got = {}
    for name, x, buf, buf_full, rows in (("b", b, bb, bb_full, m), ("q", q, qb, qb_full, n)):
        xn, _ = orc.l2norm_rows(x)
        v6, err, gerr = _mx6_rows(buf, rows, d)
        np.testing.assert_array_equal(v6[:, :h].astype(np.float64), mx6_image(xn)[:, :h])
        assert np.isinf(err).all() and np.isinf(gerr).all()
        errh, gerrh = _mx6_half_err(buf, rows, d)
        errh_full, gerrh_full = _mx6_half_err(buf_full, rows, d)
        np.testing.assert_array_equal(errh, errh_full)
        np.testing.assert_array_equal(gerrh, gerrh_full)
        true = np.linalg.norm(xn[:, :h].astype(np.float64) - v6[:, :h].astype(np.float64), axis=1)
        assert (errh >= true).all() and (errh <= true * 1.0003 + 1.1e-3).all()
        gmax = np.array([errh[g:g + 128].max() for g in range(0, rows, 128)])
        np.testing.assert_array_equal(gerrh[::128], gmax)
        _, err_full, _ = _mx6_rows(buf_full, rows, d)
        assert (errh <= err_full).all()            # never looser than the full-width E the pass used before
        got[name] = (xn, v6, errh)
    (qn, q6, eq), (bn, b6, eb) = got["q"], got["b"]
    ii, jj = rng.integers(0, n, 20000), rng.integers(0, m, 20000)
    exact = np.einsum("ij,ij->i", qn[ii, :h].astype(np.float64), bn[jj, :h].astype(np.float64))
    coarse = np.einsum("ij,ij->i", q6[ii, :h].astype(np.float64), b6[jj, :h].astype(np.float64))
    bound = (1.0 + 2.0 ** -13 + eb[jj]) * eq[ii] + (1.0 + 2.0 ** -13) * eb[jj]
    assert (np.abs(exact - coarse) <= bound).all()


def test_mx6_bound_holds_pair_by_pair():
    """| cos(a, b) - image_a . image_b | <= (1 + 2^-13 + E_b) E_a + (1 + 2^-13) E_b for sampled pairs (fp64), and for unit Gaussian
    rows it is the ~0.06 the header states."""
    d, n, m = 384, 2304, 4096
    rng = np.random.default_rng(7)
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    qb, bb = _prepare(torch.from_numpy(b).cuda(), torch.from_numpy(q).cuda())
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    q6, eq, _ = _mx6_rows(qb, n, d)
    b6, eb, _ = _mx6_rows(bb, m, d)
    ii, jj = rng.integers(0, n, 20000), rng.integers(0, m, 20000)
    exact = np.einsum("ij,ij->i", qn[ii].astype(np.float64), bn[jj].astype(np.float64))
    coarse = np.einsum("ij,ij->i", q6[ii].astype(np.float64), b6[jj].astype(np.float64))
    bound = (1.0 + 2.0 ** -13 + eb[jj]) * eq[ii] + (1.0 + 2.0 ** -13) * eb[jj]
    assert (np.abs(exact - coarse) <= bound).all()
    assert 0.04 < bound.mean() < 0.075


@pytest.mark.parametrize("d,n,m", [(384, 2500, 9000), (256, 3000, 5003), (384, 2100, 130), (384, 300, 20011), (512, 2300, 4100),
                                   (768, 2200, 3000)])
def test_mx6_pass_gives_the_oracle_answers_and_keeps_the_gate_contract(d, n, m):
    """VFM_RECORDS_MX6 on planted matches, heavy-tailed rows, rows that are all alike, exact duplicates, zero rows and
    rows of tiny norm: every resolved query has the oracle's index and similarity, every unresolved one is below the gate in the
    oracle, with the gate at 0.8 and switched off.  (384, 300, .) has no fp6 kernel: the call behaves as the int8 kinds there; d = 512
    and 768 have the half-width fp6 kernel only -- the image comes from kernels of its own -- and the full-width kinds behave as their
    int8 forms.)"""
    rng = np.random.default_rng(d + n)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    cases = {}
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    q[::3] = rng.standard_normal((len(q[::3]), d)).astype(np.float32)
    cases["planted"] = (q, b)
    cases["heavy-tailed"] = (_heavy_tailed(rng, n, d), _heavy_tailed(rng, m, d))
    base = rng.standard_normal((1, d)).astype(np.float32)
    cases["all alike"] = (base + 0.2 * rng.standard_normal((n, d)).astype(np.float32), base + 0.2 * rng.standard_normal((m, d)).astype(np.float32))
    few = rng.standard_normal((16, d)).astype(np.float32)
    cases["duplicates"] = (few[rng.integers(0, 16, n)] + 0.0, few[rng.integers(0, 16, m)] + 0.0)
    b3, q3 = b.copy(), q.copy()
    b3[100 % m] = 0.0
    b3[(m // 2):(m // 2) + 3] = 0.0
    q3[12] = 0.0
    q3[13] = 1e-18 * q3[13]
    cases["zero and tiny rows"] = (q3, b3)
    for name, (qq, bb) in cases.items():
        qn, _ = orc.l2norm_rows(qq)
        bn, _ = orc.l2norm_rows(bb)
        ridx, rsim = orc.match_ip_top1(qn, bn)
        qd, bd = torch.from_numpy(qq).cuda(), torch.from_numpy(bb).cuda()
        for g in (gate, float("-inf")):
            for records, flags in ([(RECORDS_MX6, PREPARE_MX6), (RECORDS_MX6_TOP2, PREPARE_MX6), (RECORDS_MX6_PILOT, PREPARE_MX6)] +
                                   ([(k, f) for k in HALF_KINDS for f in (PREPARE_MX6, PREPARE_MX6 | PREPARE_MX6_HALF)] if g > float("-inf") else []) +
                                   ([(RECORDS_MX6_FUSED, PREPARE_MX6)] if g > float("-inf") else [])):
                idx, sim = _search(qd, bd, g, records, flags=flags)
                solved = _gate_contract(idx, sim, ridx, rsim, g)
                if g == float("-inf"):
                    assert solved.all(), (name, records)
                assert solved[rsim >= 0.8].all(), (name, records)


def test_mx6_pipeline_mode_equals_the_oracle_registration():
    """RegistrationPipeline(coarse="mx6") -- prepare with the fp6 image, fp6 coarse pass, int8 rescans, fp64 decision, RANSAC --
    against the oracle's registration of the same pair: correspondences, pose, inlier mask, winner; serial and overlapped."""
    n, m, d = 3000, 20000, 384
    p = synth.make_pair_device(n, m, d, seed=11)
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    keep = ~(rsim.astype(np.float64) < 0.8)
    corres = np.stack([np.nonzero(keep)[0], ridx[keep]], 1).astype(np.int32)
    ref = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, 2000, seed=42)
    for overlap, mode in ((False, "mx6"), (True, "mx6"), (True, "mx6-top2"), (True, "mx6-half"), (False, "mx6-half"), (True, "mx6-fused"),
                          (False, "mx6-fused")):
        pipe = RegistrationPipeline(n, m, d, n_iter=2000, overlap_ransac=overlap, overlap_prepare=overlap, solve_streams=2, coarse=mode)
        for _ in range(3):
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        pipe.synchronize()
        torch.cuda.synchronize()
        c = int(out["count"].item())
        assert c == len(corres)
        np.testing.assert_array_equal(out["corres"].cpu().numpy()[:c], corres)
        np.testing.assert_array_equal(out["T"].cpu().numpy(), ref.transformation)
        assert int(out["best_hyp"].item()) == ref.best_hyp
        np.testing.assert_array_equal(out["mask"].cpu().numpy()[:c], ref.inlier_mask[:c])


def soak_trial_mx6(lib, rng, st):
    """one random trial of the fp6 pass (VFM_RECORDS_MX6 on operands prepared with VFM_PREPARE_MX6) against best-score int8
    records: random shape (below and above the 2048 queries the fp6 kernel starts at), width, gate (or none) and data kind.
    Gate contract, pairwise: the same answer where both resolve; whatever only one resolves lies below the gate; the matches
    a caller keeps are identical; without a gate both resolve everything."""
    d = int(rng.choice([256, 384, 384, 512, 768]))
    n = int(rng.integers(1, 9000))
    m = int(rng.integers(1, 60000))
    gsel = rng.choice([0.5, 0.8, 0.8, 0.95, -1.0])
    gate = float("-inf") if gsel < 0 else float(np.nextafter(np.float32(gsel), np.float32(-np.inf)))
    kind = rng.choice(["planted", "alike", "duplicates", "scaled"])
    g = torch.Generator(device="cuda")
    g.manual_seed(int(rng.integers(1 << 30)))
    b = torch.randn((m, d), generator=g, device="cuda")
    pick = torch.randint(0, m, (n,), generator=g, device="cuda")
    q = b[pick] + float(rng.choice([0.1, 0.3, 0.6])) * torch.randn((n, d), generator=g, device="cuda")
    if kind == "alike":
        base = torch.randn((1, d), generator=g, device="cuda")
        b = base + 0.3 * b
        q = base + 0.3 * q
    elif kind == "duplicates":
        b = b[torch.randint(0, max(1, m // 50), (m,), generator=g, device="cuda")].clone()
        q = b[pick].clone()
    elif kind == "scaled":   # blocks of very different magnitude inside a row: the block scales do the work
        b[:, : d // 4] *= 1e-3
        b[:, d // 2:] *= 30.0
        q[:, : d // 4] *= 1e-3
        q[:, d // 2:] *= 30.0
    q[torch.rand(n, generator=g, device="cuda") < 0.3] = torch.randn((d,), generator=g, device="cuda")
    q, b = q.contiguous(), b.contiguous()
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, PREPARE_MX6, st))
    res = {}
    for records in (0, RECORDS_MX6, RECORDS_MX6_TOP2, RECORDS_MX6_PILOT) + (HALF_KINDS + (RECORDS_MX6_FUSED,) if gate > float("-inf") else ()):
        idx = torch.empty(n, dtype=torch.int64, device="cuda")
        sim = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
        torch.cuda.synchronize()
        res[records] = (idx, sim)
    (i0, s0) = res[0]
    ok = True
    for records in (RECORDS_MX6, RECORDS_MX6_TOP2, RECORDS_MX6_PILOT) + HALF_KINDS + (RECORDS_MX6_FUSED,):
        if records not in res:
            continue
        i, s = res[records]
        both = (i >= 0) & (i0 >= 0)
        ok &= bool(torch.equal(i[both], i0[both]) and torch.equal(s[both], s0[both]))
        ok &= bool((s0[(i0 >= 0) & (i < 0)] < gate).all()) and bool((s[(i >= 0) & (i0 < 0)] < gate).all())
        keep, keep0 = s >= gate, s0 >= gate
        ok &= bool(torch.equal(keep, keep0) and torch.equal(i[keep], i0[keep0]))
        if gate == float("-inf"):
            ok &= bool((i >= 0).all() and (i0 >= 0).all())
    return ok, (f"d {d} n {n} m {m} gate {gate:.3f} {kind}: kept {int((s0 >= gate).sum())}, resolved int8 / fp6 / fp6 top-2 "
                f"{int((i0 >= 0).sum())}/{int((res[RECORDS_MX6][0] >= 0).sum())}/{int((res[RECORDS_MX6_TOP2][0] >= 0).sum())}")


def test_mx6_pass_randomised_soak_fixed_seed():
    lib = _lib.load()
    rng = np.random.default_rng(20260930)
    st = torch.cuda.current_stream().cuda_stream
    bad = []
    for t in range(12):
        ok, desc = soak_trial_mx6(lib, rng, st)
        if not ok:
            bad.append(desc)
    assert not bad, bad


def test_fp6_record_kinds_on_operands_without_the_fp6_image_stay_correct_and_half_width_kinds_work_with_it():
    """Misuse is safe: an operand prepared WITHOUT VFM_PREPARE_MX6 carries infinite fp6 bounds, so an fp6 record kind prunes
    nothing on it and the exact decision (all-pairs fallback, or the guard's full-width pass for the half-width kind) gives the
    oracle's answers; and an operand prepared WITH the flag still carries the int8 half-width image for VFM_RECORDS_HALF / _FUSED."""
    d, n, m = 384, 2100, 1300
    rng = np.random.default_rng(5)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = b[rng.integers(0, m, n)] + 0.3 * rng.standard_normal((n, d)).astype(np.float32)
    q[::3] = rng.standard_normal((len(q[::3]), d)).astype(np.float32)
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    ridx, rsim = orc.match_ip_top1(qn, bn)
    qd, bd = torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda()
    for records in (RECORDS_MX6, RECORDS_MX6_TOP2) + HALF_KINDS:
        idx, sim = _search(qd, bd, gate, records, flags=0)          # no fp6 image
        solved = _gate_contract(idx, sim, ridx, rsim, gate)
        assert solved[rsim >= 0.8].all(), records
    for records in (3, 4):
        idx, sim = _search(qd, bd, gate, records, flags=PREPARE_MX6)   # fp6 image AND the int8 half-width one
        solved = _gate_contract(idx, sim, ridx, rsim, gate)
        assert solved[rsim >= 0.8].all(), records


def test_a_map_prepared_once_keeps_the_pipeline_off_the_fp6_kinds():
    """prepare_map() + register(reuse_map=True) -- the IndexFlatIP.add-once form: the map operand then carries no fp6 image, so
    `auto` must settle on the int8 half-width pass, not the fp6 one, and give the oracle's registration; the pinned fp6 modes
    refuse prepare_map()."""
    n, m, d = 3000, 20000, 384
    p = synth.make_pair_device(n, m, d, seed=13)
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    keep = ~(rsim.astype(np.float64) < 0.8)
    corres = np.stack([np.nonzero(keep)[0], ridx[keep]], 1).astype(np.int32)
    ref = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, 2000, seed=42)
    pipe = RegistrationPipeline(n, m, d, n_iter=2000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="auto")
    pipe.prepare_map(p["b_desc"])
    for _ in range(5):
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], reuse_map=True)
        pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
    assert pipe.half and not pipe.mx6_half and not pipe.mx6
    c = int(out["count"].item())
    assert c == len(corres)
    np.testing.assert_array_equal(out["corres"].cpu().numpy()[:c], corres)
    np.testing.assert_array_equal(out["T"].cpu().numpy(), ref.transformation)
    with pytest.raises(ValueError):
        RegistrationPipeline(n, m, d, n_iter=2000, coarse="mx6-half").prepare_map(p["b_desc"])


def test_reuse_map_without_prepare_map_keeps_auto_off_the_fp6_kinds():
    """register(reuse_map=True) with no prepare_map() before it: the first registration prepares the map through
    vfm_match_prepare2, which writes no fp6 image -- `auto` must never pick record kinds 5 / 6 / 7 on such operands (their err6 is
    unset: nothing would be pruned), on D.2 data (where it would otherwise settle on kind 7) and on lifted-like data (kind 5);
    pinned fp6 modes refuse reuse_map; the registrations equal the oracle's."""
    n, m, d = 3000, 20000, 384
    for name, p in (("D.2", synth.make_pair_device(n, m, d, seed=14)),
                    ("lifted", synth.make_lifted_pair_device(n, m, d, seed=15, clouds=4, view_noise=0.1, common=0.0))):
        qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
        bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
        ridx, rsim = orc.match_ip_top1(qn, bn)
        keep = ~(rsim.astype(np.float64) < 0.8)
        corres = np.stack([np.nonzero(keep)[0], ridx[keep]], 1).astype(np.int32)
        ref = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, 2000, seed=42)
        pipe = RegistrationPipeline(n, m, d, n_iter=2000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="auto")
        kinds = []
        for _ in range(8):
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], reuse_map=True)
            kinds.append(pipe._records())
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
        assert not any(k in (5, 6, 7, 8, 9) for k in kinds + [pipe._records()]), (name, kinds)
        assert not pipe.mx6 and not pipe.mx6_half, name
        c = int(out["count"].item())
        assert c == len(corres), name
        np.testing.assert_array_equal(out["corres"].cpu().numpy()[:c], corres, err_msg=name)
        np.testing.assert_array_equal(out["T"].cpu().numpy(), ref.transformation, err_msg=name)
        # ... and back: the fp6 passes are off only while a reused map is searched -- registrations that prepare map
        # and scan together again get them back (D.2: the half-width pass on the fp6 image, record kind 8), same answers
        kinds2 = []
        for _ in range(8):
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
            kinds2.append(pipe._records())
            pipe.synchronize()
            torch.cuda.synchronize()
            pipe._poll_feedback()
        assert pipe._mx6_half_ok, name
        if name == "D.2":
            assert kinds2[-1] == 8 and pipe.mx6_half, (name, kinds2)
        assert int(out["count"].item()) == c, name
        np.testing.assert_array_equal(out["corres"].cpu().numpy()[:c], corres, err_msg=name)
        np.testing.assert_array_equal(out["T"].cpu().numpy(), ref.transformation, err_msg=name)
        del pipe
    for coarse in ("mx6", "mx6-top2", "mx6-half"):
        with pytest.raises(ValueError):
            RegistrationPipeline(n, m, d, n_iter=2000, coarse=coarse).register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], reuse_map=True)


@pytest.mark.parametrize("records", [0, RECORDS_MX6])
def test_finish_stage_staging_buffers_overflow_into_the_direct_paths(records):
    """More candidates than the selection stages per query tile (> 192 candidate chunks per query on a 313-chunk map: rows that are
    all alike) and more hit rows than a rescan workgroup stages (> 2048 per chunk slice: a few distinct rows repeated through the
    whole map): match_select_best_kernel places the rest on the spot (looking at the bin counters first), match_rescan_chunk_kernel
    appends the rest directly, lists that outgrow their capacity go to the all-pairs kernel -- oracle answers and gate contract,
    int8 and fp6 best-score records, gate on and off."""
    d, n, m = 384, 2304, 40000
    rng = np.random.default_rng(records + 17)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    base = rng.standard_normal((1, d)).astype(np.float32)
    alike = (base + 0.2 * rng.standard_normal((n, d)).astype(np.float32), base + 0.2 * rng.standard_normal((m, d)).astype(np.float32))
    few = rng.standard_normal((24, d)).astype(np.float32)
    dup_b = few[rng.integers(0, 24, m)] + 0.0
    dup_q = few[rng.integers(0, 24, n)] + 0.001 * rng.standard_normal((n, d)).astype(np.float32)
    for name, (qq, bb) in {"all alike": alike, "repeated rows": (dup_q, dup_b)}.items():
        qn, _ = orc.l2norm_rows(qq)
        bn, _ = orc.l2norm_rows(bb)
        ridx, rsim = orc.match_ip_top1(qn, bn)
        qd, bd = torch.from_numpy(qq).cuda(), torch.from_numpy(bb).cuda()
        for g in (gate, float("-inf")):
            idx, sim = _search(qd, bd, g, records)
            solved = _gate_contract(idx, sim, ridx, rsim, g)
            if g == float("-inf"):
                assert solved.all(), name
            assert solved[rsim >= 0.8].all(), name


def test_crowded_lists_at_c2_size_every_record_kind_gives_the_oracle_answers():
    """C2 size on descriptors that look like lifted ViT features with a common component (bench.py's `C2_lifted`: 12 candidate chunks per
    query behind the int8 pass, 46 behind the fp6 pass, ~10 000 queries above the gate): the best-score selection (a tile's candidates
    counted per chunk in the LDS, one bin atomic per (tile, chunk)), the chunk-major rescan and the refinement as one wave per list entry
    against the oracle on all 20 000 rows; the packed top-2 kinds -- another selection kernel, query-major rescans -- must agree too."""
    n, m, d = 20000, 200000, 384
    p = synth.make_lifted_pair_device(n, m, d, seed=42, device="cuda", clouds=10, view_noise=0.1, common=1.0)
    q, b = p["q_desc"], p["b_desc"]
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    qn, _ = orc.l2norm_rows(q.cpu().numpy())
    bn, _ = orc.l2norm_rows(b.cpu().numpy())
    ridx, rsim = orc.match_ip_top1(qn, bn)
    assert int((rsim >= 0.8).sum()) > 5000
    first = None
    # (RECORDS_MX6_FUSED on these data: thousands of (query, chunk) pairs reach the gate per workgroup -- the lists overflow, the guard
    # goes up and match_gatepass_kernel decides every query: the same answers, the slow way)
    for records in (0, RECORDS_MX6, 1, RECORDS_MX6_TOP2, RECORDS_MX6_PILOT, RECORDS_MX6_FUSED):
        idx, sim = _search(q, b, gate, records)
        solved = _gate_contract(idx, sim, ridx, rsim, gate)
        assert solved[rsim >= 0.8].all(), records
        keep = (sim >= 0.8)
        if first is None:
            first = (idx[keep].clone(), keep.clone())
        else:
            assert torch.equal(keep, first[1]) and torch.equal(idx[keep], first[0]), records


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,m,flags", [(384, 3000, 9001, PREPARE_MX6), (384, 3000, 9001, PREPARE_MX6 | PREPARE_MX6_HALF),
                                         (256, 1234, 5000, PREPARE_MX6), (256, 129, 5000, PREPARE_MX6 | PREPARE_MX6_HALF),
                                         (384, 1, 127, PREPARE_MX6 | PREPARE_MX6_HALF)])
def test_the_two_forms_of_the_fp6_preparation_write_the_same_bytes(d, n, m, flags):
    """prep_stream_kernel (two passes over the rows, 4 waves: the default, fits beside a coarse workgroup) against prep_chunk_kernel
    (rows in registers): every image, every E, every group figure byte for byte -- ragged sizes, a zero row, a row with a NaN, a row
    of huge magnitude."""
    lib = _lib.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(d + n + m + flags)
    q = torch.randn((n, d), generator=g, device="cuda")
    b = torch.randn((m, d), generator=g, device="cuda")
    b[m // 2] = 0.0
    b[m // 3, 5] = float("nan")
    b[m // 4] *= 1e30
    q[0, : d // 2] *= 1e-3
    st = torch.cuda.current_stream().cuda_stream
    out = []
    try:
        for variant in (40, 41):
            lib.vfm_debug_set_coarse_variant(variant)
            qb = torch.zeros(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
            bb = torch.zeros(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
            _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st))
            torch.cuda.synchronize()
            out.append((qb, bb))
    finally:
        lib.vfm_debug_set_coarse_variant(43)     # the default (round 6): prep_once_kernel
    for k, name in ((0, "scan"), (1, "map")):
        diff = torch.nonzero(out[0][k] != out[1][k]).flatten()
        assert diff.numel() == 0, f"{name}: {diff.numel()} bytes differ, first at {diff[:8].tolist()}, last at {int(diff[-1])} of {out[0][k].numel()}"



def _i8_rows(buf, rows, d):
    lib = _lib.load()
    q8 = np.empty((rows, d), np.int8)
    step, err, gerr = (np.empty(rows, np.float32) for _ in range(3))
    _lib.check(lib.vfm_debug_i8_rows(buf.data_ptr(), rows, d, q8.ctypes.data, step.ctypes.data, err.ctypes.data, gerr.ctypes.data))
    return q8, step, err, gerr


@pytest.mark.gpu
@pytest.mark.parametrize("d,n,m,flags", [(384, 3000, 9001, PREPARE_MX6), (384, 3000, 9001, PREPARE_MX6 | PREPARE_MX6_HALF),
                                         (256, 1234, 5000, PREPARE_MX6), (256, 129, 5000, PREPARE_MX6 | PREPARE_MX6_HALF),
                                         (384, 1, 127, PREPARE_MX6 | PREPARE_MX6_HALF)])
def test_the_one_read_form_of_the_fp6_preparation(d, n, m, flags):
    """prep_once_kernel (round 6, the default: every row read ONCE, the wave's tile kept as packed halves in registers between the pass
    that finds 1 / |row| and the group's step and the pass that quantises) against prep_stream_kernel.  Identical, byte for byte:
    1 / |row| (the oracle's), the fp6 image with its block scales, err6 / err6h and their group maxima, the int8 steps.  Its own
    definition: the int8 codes are rint(fp16(v) / s) -- at most one unit from rint(v / s) -- and E(int8) = |fp16(v) - s q| measured +
    the fp16 rounding bounded; E must still bound the residual against the ORACLE's normalised rows (that is all the search's proofs
    use: tests/test_gpu_int8.py::test_quantisation_bound_holds_for_every_pair), and stay within 6 % of the other form's."""
    from oracle import oracle as orc
    lib = _lib.load()
    g = torch.Generator(device="cuda")
    g.manual_seed(d + n + m + flags)
    q = torch.randn((n, d), generator=g, device="cuda")
    b = torch.randn((m, d), generator=g, device="cuda")
    b[m // 2] = 0.0
    b[m // 4] *= 1e18
    q[0, : d // 2] *= 1e-3
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for variant in (41, 43):
            lib.vfm_debug_set_coarse_variant(variant)
            qb = torch.zeros(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
            bb = torch.zeros(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
            _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st))
            torch.cuda.synchronize()
            out[variant] = (qb, bb)
    finally:
        lib.vfm_debug_set_coarse_variant(43)
    for k, (x, rows) in enumerate(((q, n), (b, m))):
        A, B = out[41][k], out[43][k]
        assert torch.equal(A[: 4 * rows], B[: 4 * rows])                               # 1 / |row|
        for u, v in zip(_mx6_rows(A, rows, d), _mx6_rows(B, rows, d)):                 # fp6 image (dequantised), err6, gerr6
            np.testing.assert_array_equal(u, v)
        eh = [np.empty(rows, np.float32) for _ in range(4)]
        _lib.check(lib.vfm_debug_mx6_half_err(A.data_ptr(), rows, d, eh[0].ctypes.data, eh[1].ctypes.data))
        _lib.check(lib.vfm_debug_mx6_half_err(B.data_ptr(), rows, d, eh[2].ctypes.data, eh[3].ctypes.data))
        np.testing.assert_array_equal(eh[0], eh[2])
        np.testing.assert_array_equal(eh[1], eh[3])
        a8, b8 = _i8_rows(A, rows, d), _i8_rows(B, rows, d)
        np.testing.assert_array_equal(a8[1], b8[1])                                    # the groups' steps
        assert np.abs(a8[0].astype(np.int32) - b8[0].astype(np.int32)).max() <= 1
        v, _ = orc.l2norm_rows(x.cpu().numpy())
        res = np.linalg.norm(v.astype(np.float64) - b8[1][:, None].astype(np.float64) * b8[0].astype(np.float64), axis=1)
        assert (b8[2].astype(np.float64) >= res).all()
        assert (b8[3] >= b8[2]).all()
        ok = a8[2] > 0
        assert (b8[2][ok] <= 1.06 * a8[2][ok] + 5e-4).all()


@pytest.mark.gpu
@pytest.mark.parametrize("data", ["D.2", "lifted"])
def test_searches_behind_either_form_of_the_preparation_give_the_same_answers(data):
    """the exact decision does not depend on which int8 image the rescans read: fp6 half-width fused, fp6 full width, int8 records"""
    from vfmreg import synth
    lib = _lib.load()
    n, m, d = 6000, 40000, 384
    p = synth.make_pair_device(n, m, d, seed=3) if data == "D.2" else synth.make_lifted_pair_device(n, m, d, seed=3, common=1.0)
    gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
    res = {}
    try:
        for variant in (41, 43):
            lib.vfm_debug_set_coarse_variant(variant)
            for records, flags in ((8, PREPARE_MX6 | PREPARE_MX6_HALF), (5, PREPARE_MX6), (0, PREPARE_MX6)):
                res[(variant, records)] = _search(p["q_desc"], p["b_desc"], gate, records, flags)
    finally:
        lib.vfm_debug_set_coarse_variant(43)
    for records in (8, 5, 0):
        a, b_ = res[(41, records)], res[(43, records)]
        assert torch.equal(a[0], b_[0]) and torch.equal(a[1], b_[1]), records
