#!/usr/bin/env python
"""Several builds of the library in ONE process, their operand preparation at C2 (flags 24, the headline's) timed alternately -- see
tools/ab_two_libs_coarse.py.  Also checks that the builds write the same bytes.   python tools/ab_two_libs_prep.py libA.so libB.so [...]"""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

base = _lib.load()
libs = []
for name in sys.argv[1:]:
    l = C.CDLL(str(ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / name))
    l.vfm_match_prepare2_gated_p.restype = C.c_int
    l.vfm_match_prepare2_gated_p.argtypes = _lib.SIGNATURES["vfm_match_prepare2_gated_p"][1]
    libs.append((name, l))
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
st = torch.cuda.current_stream().cuda_stream
bufs = {}
for name, _ in libs:
    bufs[name] = (torch.zeros(base.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda"),
                  torch.zeros(base.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda"))
acc = {name: [] for name, _ in libs}
for rep in range(12):
    for name, l in libs:
        qb, bb = bufs[name]
        call = lambda: l.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24, st)
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            acc[name].append(e0.elapsed_time(e1) / 20)
ref = bufs[libs[0][0]]
for name, _ in libs:
    v = sorted(acc[name])
    same = all(bool(torch.equal(a, b)) for a, b in zip(ref, bufs[name]))
    print(f"{name:34s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}   same bytes as {libs[0][0]}: {same}", flush=True)
