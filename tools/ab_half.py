#!/usr/bin/env python
"""Half-width pass (VFM_RECORDS_HALF = 3; 4 = the same with the selection fused into the coarse kernel) against best-score records (0) and packed top-2 records (1) of the gated family:
coarse-kernel time, finish time, surviving chunks, and agreement of the answers under the gate contract (a query resolved by
both has the same index and similarity; a query left unresolved by either has a best-score-records similarity below the gate)."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

lib = _lib.load()
n, m, d = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (20000, 200000, 384)))
p = synth.make_pair_device(n, m, d, seed=42)
q, b = p["q_desc"], p["b_desc"]
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.vfm_match_prepare2_gated(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, st))
gate = float(os.environ.get("VFM_GATE", "0.7999999"))
a, e = C.c_void_p(), C.c_void_p()
_lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(e)))
ms = C.c_float()
res = {}
for records in tuple(int(v) for v in os.environ.get("VFM_AB_RECORDS", "0,3,4,1,0,3,4").split(",")):
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    tc, tf = [], []
    for i in range(10):
        lib.vfm_prof_arm(a, e)
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
        torch.cuda.synchronize()
        _lib.check(lib.vfm_prof_elapsed_ms(a, e, C.byref(ms)))
        t0 = time.perf_counter()
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
        torch.cuda.synchronize()
        if i >= 2:
            tc.append(ms.value)
            tf.append(1e3 * (time.perf_counter() - t0))
    lib.vfm_debug_set_match_stats(1)
    _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
    _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                   sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
    stats = (C.c_int32 * 64)()
    _lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, stats))
    lib.vfm_debug_set_match_stats(0)
    s = list(stats)
    print(f"records {records}: coarse {sorted(tc)[len(tc) // 2]:.3f} ms, finish {sorted(tf)[len(tf) // 2]:.3f} ms; resolved {int((idx >= 0).sum())}, "
          f"fallbacks {s[0]}, refined {s[1]}, candidate chunks/query {s[2] / n:.3f}, rescans {s[5]}, crowded {s[6]}", flush=True)
    res.setdefault(records, (idx.clone(), sim.clone()))
ref = res.get(0)
if ref is not None:
    for r, (idx, sim) in res.items():
        both = (idx >= 0) & (ref[0] >= 0)
        only_ref = (ref[0] >= 0) & (idx < 0)
        only_new = (idx >= 0) & (ref[0] < 0)
        ok = bool(torch.equal(idx[both], ref[0][both]) and torch.equal(sim[both], ref[1][both]))
        below = bool((ref[1][only_ref] < gate).all())
        print(f"records {r} vs 0: same answer where both resolve {ok}; {int(only_ref.sum())} queries only records 0 resolves (all below the gate: "
              f"{below}); {int(only_new.sum())} queries only records {r} resolves; matches >= 0.8: {int((sim >= 0.8).sum())} vs {int((ref[1] >= 0.8).sum())}")
