#!/usr/bin/env python
"""The finish stage of a gated search on bench.py's lifted descriptors, kernel by kernel (run under rocprofv3 --kernel-trace by
tools/prof_finish.sh): record kinds $1 (comma list, default 0,5), select variants $2 (vfm_debug_set_coarse_variant values)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
kinds = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,5").split(",")]
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
data = sys.argv[3] if len(sys.argv) > 3 else "lifted"
p = (synth.make_lifted_pair_device(n, m, d, seed=42, device="cuda", clouds=10, view_noise=0.1, common=1.0) if data == "lifted"
     else synth.make_pair_device(n, m, d, seed=42))
q, b = p["q_desc"], p["b_desc"]
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, 8, st))
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
idx = torch.empty(n, dtype=torch.int64, device="cuda")
sim = torch.empty(n, dtype=torch.float32, device="cuda")
gate = 0.8
for rec in kinds:
    for v in variants:
        lib.vfm_debug_set_coarse_variant(v)
        for rep in range(4):
            _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), rec, gate, st))
            _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                           sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, rec, st))
            torch.cuda.synchronize()
        print("MARK", rec, v, int((idx >= 0).sum()), flush=True)
lib.vfm_debug_set_coarse_variant(0)
