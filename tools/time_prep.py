#!/usr/bin/env python
"""Operand preparation alone (map + scan in one launch): time per call at C2."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=1)
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for fn, label in ((lib.vfm_match_prepare2_gated, "gated family: int8 image only"), (lib.vfm_match_prepare2, "ungated: int8 + fp16 images")):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.check(fn(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, st))
        torch.cuda.synchronize()
        print(f"{label}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per call", flush=True)
