"""Kernel policy as a caller-owned object (round 6; VERDICT r5 item 7, SURVEY.md 8 B.5 "no global state except the last-error string"):
include/vfmreg.h vfm_config_*.  A vfm_config_t is bound per THREAD; two pipelines with different kernel variants run from two threads
at once, each gets the kernels ITS config names, and both return the oracle's registration."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import oracle as orc                   # noqa: E402
from vfmreg import _lib, synth                     # noqa: E402
from vfmreg.pipeline import RegistrationPipeline   # noqa: E402


def _err_of_prepared(buf, rows, d):
    lib = _lib.load()
    q8 = np.empty((rows, d), np.int8)
    step, err, gerr = (np.empty(rows, np.float32) for _ in range(3))
    _lib.check(lib.vfm_debug_i8_rows(buf.data_ptr(), rows, d, q8.ctypes.data, step.ctypes.data, err.ctypes.data, gerr.ctypes.data))
    return q8, err


def test_a_binding_is_the_calling_threads_only():
    lib = _lib.load()
    a = _lib.Config(coarse_variant=41, coarse_slices=33)
    seen = {}

    def read(name):
        v = C.c_int64()
        _lib.check(lib.vfm_config_get(None, b"prep_form", C.byref(v)))
        w = C.c_int64()
        _lib.check(lib.vfm_config_get(None, b"coarse_slices", C.byref(w)))
        seen[name] = (v.value, w.value)

    def other():
        read("other thread, nothing bound")
        with _lib.using(_lib.Config(coarse_variant=40)):
            read("other thread, its own")
        read("other thread, after the block")

    with _lib.using(a):
        read("this thread, bound")
        t = threading.Thread(target=other)
        t.start()
        t.join()
        read("this thread, still bound")
    with _lib.using(None):
        read("this thread, factory")
    assert seen["this thread, bound"] == (1, 33) and seen["this thread, still bound"] == (1, 33)
    assert seen["other thread, nothing bound"] == (3, 0) and seen["other thread, its own"] == (0, 0)
    assert seen["other thread, after the block"] == (3, 0) and seen["this thread, factory"] == (3, 0)
    assert lib.vfm_config_set(a._h, b"no such key", 1) != 0 and b"unknown key" in lib.vfm_last_error()


def test_two_pipelines_with_different_variants_run_concurrently():
    """Pipeline A: operands by prep_stream_kernel (rows read twice), 37 map slices, two query tiles per wave; pipeline B: prep_once_kernel
    (the default), the launcher's slices, three tiles.  Driven from two threads at the same time, 12 registrations each.  Which
    preparation ran is visible in the prepared operand: the two forms define E(int8) differently (tests/test_gpu_mx6.py), so the buffer
    sets of A must hold the stream form's E and those of B the one-read form's -- and every pose / correspondence set is the oracle's."""
    n, m, d, iters = 6000, 40000, 384, 3000
    pairs = [synth.make_pair_device(n, m, d, seed=70 + k) for k in range(2)]
    cfg_a = _lib.Config(coarse_variant=41, coarse_slices=37)
    cfg_a.set("coarse_variant", 32)
    cfg_b = _lib.Config()
    pipes = {name: RegistrationPipeline(n, m, d, n_iter=iters, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="mx6-half",
                                        private_streams=True, config=cfg)
             for name, cfg in (("A", cfg_a), ("B", cfg_b))}
    outs, errs = {}, []
    start = threading.Barrier(2)

    def drive(name):
        try:
            pipe = pipes[name]
            st = torch.cuda.Stream()
            res = []
            with torch.cuda.stream(st):
                start.wait()
                for i in range(12):
                    p = pairs[i % 2]
                    out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
                    with torch.cuda.stream(out["result_stream"]):
                        res.append((out["T"].clone(), out["count"].clone(), out["corres"].clone()))
                pipe.synchronize()
            st.synchronize()
            outs[name] = res
        except Exception as e:   # (surfaced in the main thread)
            errs.append((name, e))

    ts = [threading.Thread(target=drive, args=(k,)) for k in ("A", "B")]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    # the oracle's registrations
    refs = []
    for p in pairs:
        qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
        bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
        ridx, rsim = orc.match_ip_top1(qn, bn)
        keep = ~(rsim.astype(np.float64) < 0.8)
        corres = np.stack([np.nonzero(keep)[0], ridx[keep]], 1).astype(np.int32)
        refs.append((corres, orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, iters, seed=42)))
    for name in ("A", "B"):
        for i, (T, c, corres) in enumerate(outs[name]):
            rc, rr = refs[i % 2]
            k = int(c.item())
            assert k == len(rc), (name, i)
            np.testing.assert_array_equal(corres[:k].cpu().numpy(), rc)
            np.testing.assert_array_equal(T.cpu().numpy(), rr.transformation)
    # which preparation each pipeline's calls took: E(int8) of its last map operand against both forms run alone
    lib = _lib.load()
    want = {}
    for variant in (41, 43):
        with _lib.using(_lib.Config(coarse_variant=variant)):
            bb = torch.zeros(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
            qb = torch.zeros(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
            p = pairs[1]                                   # registration 11 used pair 1
            _lib.check(lib.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24,
                                                      torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            want[variant] = _err_of_prepared(bb, m, d)[1]
    assert not np.array_equal(want[41], want[43])
    for name, variant in (("A", 41), ("B", 43)):
        r = pipes[name].sets[11 % len(pipes[name].sets)]   # the buffer set of registration 11
        got = _err_of_prepared(r.bprep, m, d)[1]
        np.testing.assert_array_equal(got, want[variant], err_msg=f"pipeline {name} did not prepare its operands with ITS config's kernel")
