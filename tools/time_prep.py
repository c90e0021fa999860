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
import hashlib
ref = {}
for grid in (-1, 0, 512, 0, -1):   # -1 = one workgroup per 128-row group (default), 0 = one per compute unit, 512 = two per compute unit
  lib.vfm_debug_set_prep_grid(grid)
  for fn, label in ((lib.vfm_match_prepare2_gated, "gated family: int8 image only"), (lib.vfm_match_prepare2, "ungated: int8 + fp16 images")):
    qb.zero_(); bb.zero_()
    for rep in range(1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            _lib.check(fn(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, st))
        torch.cuda.synchronize()
        el = 1e3 * (time.perf_counter() - t0) / 20
        h = hashlib.sha1(bb.cpu().numpy().tobytes() + qb.cpu().numpy().tobytes()).hexdigest()[:12]
        same = ref.setdefault(label, h) == h
        print(f"grid {grid:4d} {label}: {el:.3f} ms per call   prepared bytes identical to the first configuration: {same}", flush=True)
lib.vfm_debug_set_prep_grid(-1)
