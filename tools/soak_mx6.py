#!/usr/bin/env python
"""Randomised soak of the fp6 pass (VFM_RECORDS_MX6) against best-score int8 records: random shapes, widths, gates and data kinds
(the trial lives in tests/test_gpu_mx6.py, which runs twelve with a fixed seed in the suite).   python tools/soak_mx6.py [trials] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
from tests.test_gpu_mx6 import soak_trial_mx6  # noqa: E402
from vfmreg import _lib  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bad = 0
for t in range(trials):
    ok, desc = soak_trial_mx6(lib, rng, st)
    bad += 0 if ok else 1
    print(f"trial {t}: {desc} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{trials} trials, {bad} mismatches")
