#!/usr/bin/env python
"""Every query resolved (no gate): the fp16 pass (ungated call with variant 5 = fp16 everywhere), the ungated call as the
library routes it (int8 pass with top-2 records from 8192 queries x 1e9 pairs on), and the int8 pass with best-score / packed
top-2 records and gate = -inf.  Milliseconds for coarse + finish, answers compared."""
import ctypes as C
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
q, b = p["q_desc"], p["b_desc"]
Q, B = ops.PreparedRows(q), ops.PreparedRows(b)
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ref = None
for label, records in (("fp16 pass (ungated call, variant 5)", -5), ("ungated call, default routing", None),
                       ("int8, best-score records, gate -inf", 0), ("int8, top-2 records, gate -inf", 1)):
    lib.vfm_debug_set_coarse_variant(5 if records == -5 else 0)
    if records == -5:
        records = None
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    ts = []
    for r in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if records is None:
            _lib.check(lib.vfm_match_search_coarse(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), st))
            _lib.check(lib.vfm_match_search_finish(q.data_ptr(), Q.buf.data_ptr(), n, b.data_ptr(), B.buf.data_ptr(), m, d, idx.data_ptr(),
                                                   sim.data_ptr(), ws.data_ptr(), ws.numel(), st))
        else:
            _lib.check(lib.vfm_match_search_coarse_gated_r(Q.buf.data_ptr(), n, B.buf.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, st))
            _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), Q.buf.data_ptr(), n, b.data_ptr(), B.buf.data_ptr(), m, d,
                                                           idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), float("-inf"), records, st))
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(1e3 * (time.perf_counter() - t0))
    same = True if ref is None else bool(torch.equal(ref[0], idx) and torch.equal(ref[1], sim))
    if ref is None:
        ref = (idx.clone(), sim.clone())
    print(f"{label}: {sorted(ts)[len(ts) // 2]:.2f} ms (coarse + finish), same answers {same}", flush=True)
lib.vfm_debug_set_coarse_variant(0)
