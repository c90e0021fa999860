#!/usr/bin/env python
"""A/B of the coarse kernel alone (no side streams): sparse records (default) vs dense per-chunk records (variant 4),
and the sparse kernel with the rare path disabled (negative window: timing only)."""
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


def run(label, variant, window):
    lib.vfm_debug_set_coarse_variant(variant)
    lib.vfm_debug_set_coarse_window(C.c_float(window))
    t = []
    for i in range(12):
        lib.vfm_prof_arm(a, b)
        _lib.check(lib.vfm_match_search_coarse(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(),
                                               torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
        if i >= 2:
            t.append(ms.value)
    print(f"{label:60s} {sum(t) / len(t):.3f} ms  (min {min(t):.3f})", flush=True)


for rep in range(2):
    run("dense records (variant 4)", 4, 0.0)
    run("sparse records + seed units (default)", 0, 0.0)
    run("sparse records, no seed units (variant 7)", 7, 0.0)
    run("sparse, rare path never taken (window -1: timing only)", 0, -1.0)
    run("sparse, window 1e-4 (timing only)", 0, 1e-4)
lib.vfm_debug_set_coarse_variant(0)
lib.vfm_debug_set_coarse_window(C.c_float(0.0))
