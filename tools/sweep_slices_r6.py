#!/usr/bin/env python
"""The headline's coarse kernel ALONE (record kind 8, fp6 half width, three query tiles per wave) against the number of map slices
(vfm_debug_set_coarse_slices; 0 = the launcher's rule): workgroups = query blocks x slices, every workgroup pays its prologue
(queries into registers, the ring primed) and the grid's last round of 256 is partly empty."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
_lib.check(lib.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24, st))
slices = [int(x) for x in sys.argv[1:]] or [0, 16, 19, 24, 28, 32, 38, 40, 47, 52, 56, 60, 64, 0]
for rep in range(2):
    for s in slices:
        lib.vfm_debug_set_coarse_slices(s)
        for _ in range(5):
            _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
        e1.record()
        torch.cuda.synchronize()
        print(f"slices {s:3d}: {e0.elapsed_time(e1) / 40:.4f} ms per coarse call (memset + kernel), back to back", flush=True)
lib.vfm_debug_set_coarse_slices(0)
