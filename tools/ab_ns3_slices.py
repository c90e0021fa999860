#!/usr/bin/env python
"""Map slices of the fused fp6 half-width kernel with three query tiles per wave (variant 33), pipeline rate over 200 steps; variant 32 beside it."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + i) for i in range(4)]
for variant, slices in [(32, 0), (33, 0), (33, 32), (33, 40), (33, 48), (33, 54), (33, 64), (32, 0), (33, 0)]:
    lib.vfm_debug_set_coarse_variant(variant)
    lib.vfm_debug_set_coarse_slices(slices)
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="mx6-half")
    for i in range(12):
        pr = pairs[i % 4]
        pipe.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    rates = []
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(200):
            pr = pairs[i % 4]
            out = pipe.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
        pipe.synchronize()
        torch.cuda.synchronize()
        rates.append(200 / (time.perf_counter() - t0))
    print(f"variant {variant} slices {slices or 'rule'}: {rates[0]:.1f} / {rates[1]:.1f} registrations/s over 200 steps; pose err "
          f"{float(np.linalg.norm(out['T'].cpu().numpy() - pairs[199 % 4]['T_gt'])):.4f}", flush=True)
    del pipe
lib.vfm_debug_set_coarse_slices(0)
