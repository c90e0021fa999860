#!/usr/bin/env python
"""Timing only: the int8 coarse kernel under artificial windows (how much of its time is record emission)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, ops, synth  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
Q, B = ops.PreparedRows(p["q_desc"]), ops.PreparedRows(p["b_desc"])
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
a, b = C.c_void_p(), C.c_void_p()
_lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
ms = C.c_float()
st = torch.cuda.current_stream().cuda_stream
idx = torch.empty(n, dtype=torch.int64, device="cuda")
sim = torch.empty(n, dtype=torch.float32, device="cuda")
for variant, window in ((0, 0.0), (8, -1.0), (8, 0.01), (8, 0.02), (8, 0.03), (8, 0.037), (8, 0.045), (8, 0.0)):
    lib.vfm_debug_set_coarse_variant(variant)
    lib.vfm_debug_set_coarse_window(C.c_float(window))
    t = []
    for i in range(10):
        lib.vfm_prof_arm(a, b)
        _lib.check(lib.vfm_match_search_coarse(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
        if i >= 2:
            t.append(ms.value)
    rec = ws  # records per query: read rec_cnt through the stats path
    lib.vfm_debug_set_match_stats(1)
    _lib.check(lib.vfm_match_search_coarse(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), st))
    _lib.check(lib.vfm_match_search_finish(p["q_desc"].data_ptr(), Q.buf.data_ptr(), n, p["b_desc"].data_ptr(), B.buf.data_ptr(), m, d,
                                           idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), st))
    stats = (C.c_int32 * 64)()
    _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, stats))
    lib.vfm_debug_set_match_stats(0)
    print(f"variant {variant} window {window}: coarse {sum(t) / len(t):.3f} ms (min {min(t):.3f}), records/query {stats[4] / n:.1f}", flush=True)
lib.vfm_debug_set_coarse_variant(0)
lib.vfm_debug_set_coarse_window(C.c_float(0.0))
