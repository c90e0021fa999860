#!/usr/bin/env python
"""A/B of the coarse kernel alone (no side streams): sparse records (default) vs dense per-chunk records (variant 4).
(Round 2 also timed the sparse kernel with its rare path disabled through a window override; that switch made results wrong
and was removed from the library in round 3 -- the figures are in DESIGN.md 4.1.)"""
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


def run(label, variant):
    lib.vfm_debug_set_coarse_variant(variant)
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
    run("dense records (variant 4)", 4)
    run("sparse records + seed units (default)", 0)
    run("sparse records, no seed units (variant 7)", 7)
lib.vfm_debug_set_coarse_variant(0)
