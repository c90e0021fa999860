#!/usr/bin/env python
"""Coarse-kernel time in the real-data regime (a few hundred queries) vs the number of map slices."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, ops, synth  # noqa: E402

lib = _lib.load()
a, b = C.c_void_p(), C.c_void_p()
_lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
ms = C.c_float()
for n, m in ((300, 50000), (1500, 100000), (2000, 200000)):
    p = synth.make_pair_device(n, m, 384, seed=1)
    Q, B = ops.PreparedRows(p["q_desc"]), ops.PreparedRows(p["b_desc"])
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, 384), dtype=torch.uint8, device="cuda")
    for sl in (0, 8, 16, 24, 32, 48, 64):
        lib.vfm_debug_set_coarse_slices(sl)
        t = []
        for i in range(12):
            lib.vfm_prof_arm(a, b)
            _lib.check(lib.vfm_match_search_coarse(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, 384, ws.data_ptr(), ws.numel(),
                                                   torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
            if i >= 2:
                t.append(ms.value)
        print(f"n={n} m={m} slices={sl:2d}: {1e3 * sum(t) / len(t):7.1f} us", flush=True)
lib.vfm_debug_set_coarse_slices(0)
