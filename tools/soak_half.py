#!/usr/bin/env python
"""Randomised soak of the half-width pass (VFM_RECORDS_HALF = 3, fused = 4) against best-score records (0) of the gated family:
random shapes, widths, gates and data kinds (the trial itself lives in tests/test_gpu_bench_config.py, which runs ten trials
with a fixed seed in the suite).   python tools/soak_half.py [trials] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
from vfmreg import _lib  # noqa: E402
from tests.test_gpu_bench_config import soak_trial  # noqa: E402

lib = _lib.load()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = torch.cuda.current_stream().cuda_stream
bad = 0
for t in range(trials):
    ok, desc = soak_trial(lib, rng, st)
    bad += 0 if ok else 1
    print(f"trial {t}: {desc} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{trials} trials, {bad} mismatches")
sys.exit(1 if bad else 0)
