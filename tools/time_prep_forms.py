#!/usr/bin/env python
"""Driver for the PMC passes of the operand preparation (tools/pmc_prep.sh): the three forms at C2, flags 24 (headline) -- a few calls each."""
import sys
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
for variant in (40, 41, 43):
    with _lib.using(_lib.Config(coarse_variant=variant)):
        for _ in range(3):
            _lib.check(lib.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24, st))
        torch.cuda.synchronize()
