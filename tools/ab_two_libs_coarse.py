#!/usr/bin/env python
"""Two builds of the library in ONE process, their fp6 half-width coarse call (record kind 8, C2) timed alternately: boxes of the pool
drift by +-3 % between processes, which hides effects of that size (tools/ablate6.py runs one build per process).
python tools/ab_two_libs_coarse.py libA.so libB.so [...]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

base = _lib.load()
libs = []
for name in sys.argv[1:]:
    l = C.CDLL(str(ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / name))
    for fn in ("vfm_match_search_coarse_gated_g", "vfm_match_prepare2_gated_p"):
        getattr(l, fn).restype = C.c_int
        getattr(l, fn).argtypes = _lib.SIGNATURES[fn][1]
    libs.append((name, l))
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(base.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(base.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(base.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
assert base.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24, st) == 0
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
acc = {name: [] for name, _ in libs}
for rep in range(12):
    for name, l in libs:
        for _ in range(3):
            l.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            assert l.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st) == 0
        e1.record()
        torch.cuda.synchronize()
        if rep >= 2:
            acc[name].append(e0.elapsed_time(e1) / 20)
for name, _ in libs:
    v = sorted(acc[name])
    print(f"{name:34s} median {v[len(v) // 2]:.4f} ms   min {v[0]:.4f}   max {v[-1]:.4f}", flush=True)
