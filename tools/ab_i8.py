#!/usr/bin/env python
"""int8 coarse pass (gated family, variant 0) vs the fp16 one (variant 5): coarse-kernel time, record / candidate counts, finish time,
and equality of the answers."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, ops, synth  # noqa: E402

lib = _lib.load()
n, m, d = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (20000, 200000, 384)))
p = synth.make_pair_device(n, m, d, seed=42)
Q, B = ops.PreparedRows(p["q_desc"]), ops.PreparedRows(p["b_desc"])
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
a, b = C.c_void_p(), C.c_void_p()
_lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
ms = C.c_float()
st = torch.cuda.current_stream().cuda_stream
res = {}
lib.vfm_debug_set_coarse_slices(int(os.environ.get("VFM_SLICES", "0")))
GATE = C.c_float(float(os.environ.get("VFM_GATE", "-inf")))
import os
VARIANTS = tuple(int(v) for v in os.environ.get("VFM_AB_VARIANTS", "0,5,0,5").split(","))
for variant in VARIANTS:
    lib.vfm_debug_set_coarse_variant(variant)
    t, tf = [], []
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    for i in range(12):
        lib.vfm_prof_arm(a, b)
        _lib.check(lib.vfm_match_search_coarse_gated(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), st))
        torch.cuda.synchronize()
        _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
        t0 = time.perf_counter()
        _lib.check(lib.vfm_match_search_finish_gated(p["q_desc"].data_ptr(), Q.buf.data_ptr(), n, p["b_desc"].data_ptr(), B.buf.data_ptr(), m, d,
                                                     idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), GATE, st))
        torch.cuda.synchronize()
        if i >= 2:
            t.append(ms.value)
            tf.append(1e3 * (time.perf_counter() - t0))
    lib.vfm_debug_set_match_stats(1)
    _lib.check(lib.vfm_match_search_coarse_gated(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), st))
    _lib.check(lib.vfm_match_search_finish_gated(p["q_desc"].data_ptr(), Q.buf.data_ptr(), n, p["b_desc"].data_ptr(), B.buf.data_ptr(), m, d,
                                                 idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), GATE, st))
    stats = (C.c_int32 * 64)()
    _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, stats))
    lib.vfm_debug_set_match_stats(0)
    s = list(stats)
    print(f"variant {variant}: coarse {sum(t) / len(t):.3f} ms (min {min(t):.3f}), finish {sum(tf) / len(tf):.3f} ms; fallbacks {s[0]}, "
          f"refined {s[1]}, candidates/query {s[2] / n:.2f}, kept {s[3]}, records/query {s[4] / n:.1f}, rescans {s[5]}, crowded queries {s[6]}, histogram {s[8:24]}", flush=True)
    res[variant] = (idx.clone(), sim.clone())
lib.vfm_debug_set_coarse_variant(0)
first = VARIANTS[0]
for v in res:
    if v != first:
        ok = res[v][0] >= 0
        ok0 = res[first][0] >= 0
        both = ok & ok0
        print(f"variant {v} vs {first}: resolved sets equal {bool((ok == ok0).all())}, idx equal {bool((res[first][0][both] == res[v][0][both]).all())}, "
              f"sim equal {bool((res[first][1][both] == res[v][1][both]).all())}")
