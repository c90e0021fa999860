#!/usr/bin/env python
"""Descriptors that look like lifted ViT features (bench.py's C2_lifted data): how many map rows / 128-row chunks per query lie within
w of the query's best cosine (what a coarse pass with bounds of total width w hands to the rescans and the fp32 refinement), and the
library's own counts for the int8 and the fp6 full-width pass (candidate chunks per query, crowded queries, rows kept)."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
common = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
p = synth.make_lifted_pair_device(n, m, d, seed=42, device="cuda", clouds=10, view_noise=0.1, common=common)
q, b = p["q_desc"], p["b_desc"]
qn = torch.nn.functional.normalize(q, dim=1)
bn = torch.nn.functional.normalize(b, dim=1)
W = (0.005, 0.01, 0.02, 0.04, 0.08, 0.12, 0.2)
rows = torch.zeros(len(W), dtype=torch.float64)
chunks = torch.zeros(len(W), dtype=torch.float64)
gated = 0
mp = (m + 127) // 128 * 128
for i in range(0, n, 1000):
    s = qn[i:i + 1000] @ bn.T
    best = s.max(1, keepdim=True).values
    ok = best[:, 0] >= 0.8
    gated += int(ok.sum())
    s = torch.nn.functional.pad(s, (0, mp - m), value=-2.0)
    for k, w in enumerate(W):
        hit = (s >= torch.clamp(best - w, min=0.8 - w)) & ok[:, None]
        rows[k] += float(hit.sum())
        chunks[k] += float(hit.view(hit.shape[0], -1, 128).any(-1).sum())
print(f"common {common}: {gated} of {n} queries reach the gate; per query (all {n}):")
for k, w in enumerate(W):
    print(f"  within {w:5.3f} of the best: {rows[k] / n:8.1f} rows in {chunks[k] / n:6.1f} chunks")

PREPARE_MX6 = 8
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, PREPARE_MX6, st))
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
idx = torch.empty(n, dtype=torch.int64, device="cuda")
sim = torch.empty(n, dtype=torch.float32, device="cuda")
gate = 0.8
for name, rec in (("int8 best-score", 0), ("int8 top-2", 1), ("fp6 best-score", 5), ("fp6 top-2", 6)):
    lib.vfm_debug_set_match_stats(1)
    for rep in range(2):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        t0.record()
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), rec, gate, st))
        t1.record()
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, rec, st))
        t2.record()
        torch.cuda.synchronize()
    s = (C.c_int32 * 64)()
    _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, C.cast(s, C.c_void_p)))
    lib.vfm_debug_set_match_stats(0)
    s = list(s)
    print(f"{name}: coarse {t0.elapsed_time(t1):.3f} ms, finish {t1.elapsed_time(t2):.3f} ms; fallbacks {s[0]}, refined {s[1]}, candidates/query {s[2] / n:.2f}, "
          f"kept {s[3]}, rescans/query {s[5] / n:.2f}, crowded queries {s[6]}, histogram {s[8:24]}; resolved {int((idx >= 0).sum())}")
