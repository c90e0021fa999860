#!/usr/bin/env python
"""A/B of the fused fp6 half-width coarse kernel at C2 size with two / three 32-query tiles per wave (vfm_debug_set_coarse_variant(32 / 33)):
same answers (against each other and against best-score int8 records), the kernel's duration alone, and the pipeline's rate."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
q, b = p["q_desc"], p["b_desc"]
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, 8 | 16, st))
a, e = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(e))
ms = C.c_float()
res = {}
for variant in (32, 33, 32, 33):
    lib.vfm_debug_set_coarse_variant(variant)
    ts = []
    for rep in range(8):
        idx = torch.empty(n, dtype=torch.int64, device="cuda")
        sim = torch.empty(n, dtype=torch.float32, device="cuda")
        lib.vfm_prof_arm(a, e)
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(), sim.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), gate, 8, st))
        torch.cuda.synchronize()
        lib.vfm_prof_elapsed_ms(a, e, C.byref(ms))
        if rep >= 2:
            ts.append(ms.value)
    res[variant] = (idx.clone(), sim.clone())
    print(f"variant {variant}: coarse kernel alone {sorted(ts)[len(ts) // 2]:.4f} ms (min {min(ts):.4f}); kept {int((sim >= 0.8).sum())}", flush=True)
print("same answers:", bool(torch.equal(res[32][0], res[33][0]) and torch.equal(res[32][1], res[33][1])))
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, 0, st))
idx0 = torch.empty(n, dtype=torch.int64, device="cuda")
sim0 = torch.empty(n, dtype=torch.float32, device="cuda")
_lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 0, gate, st))
_lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx0.data_ptr(), sim0.data_ptr(), ws.data_ptr(), ws.numel(), gate, 0, st))
torch.cuda.synchronize()
keep0 = sim0 >= 0.8
for v in (32, 33):
    keep = res[v][1] >= 0.8
    print(f"variant {v} vs int8 best-score records: same kept set {bool(torch.equal(keep, keep0))}, same indices {bool(torch.equal(res[v][0][keep], idx0[keep0]))}")
# the pipeline (bench.py's construction), 200 steps
pairs = [synth.make_pair_device(n, m, d, seed=42 + i) for i in range(4)]
for variant in (32, 33, 32, 33):
    lib.vfm_debug_set_coarse_variant(variant)
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="mx6-half")
    for i in range(12):
        pr = pairs[i % 4]
        pipe.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        pr = pairs[i % 4]
        out = pipe.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"variant {variant}: pipeline {200 / dt:.1f} registrations/s over 200 steps; pose err {float(np.linalg.norm(out['T'].cpu().numpy() - pairs[199 % 4]['T_gt'])):.4f}", flush=True)
    del pipe
