#!/usr/bin/env python
"""Timing experiment (WRONG RESULTS): the LDS-tiled residual GEMMs (proj, fc2) of an 84-image forward without their epilogue's loads of the
fp32 residual stream (`vit_hot_a` bit 1), its stores (bit 2), the fp16 copy (bit 3), with a hot token operand (bit 0)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "vfm-registration_amd"))
import numpy as np, torch
from vfmreg import _lib, vit as V
rng = np.random.default_rng(0)
model = V.ViTS14(V.random_weights(0), 1200, 1600)
imgs = torch.from_numpy(rng.integers(1, 255, (84, 1200, 1600, 3), dtype=np.uint8)).cuda()
res = {}
for rep in range(3):
    for bits in (0, 2, 4, 8, 6, 14, 1, 15):
        with _lib.using(_lib.Config().set("vit_hot_a", bits)):
            for _ in range(3): model.forward(imgs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): model.forward(imgs)
            torch.cuda.synchronize()
            res.setdefault(bits, []).append((time.perf_counter() - t0) / 10 * 1e3)
for bits, v in res.items():
    print(f"vit_hot_a {bits:2d}: {sorted(v)[1]:.3f} ms  ({(sorted(v)[1] - sorted(res[0])[1]) / 24 * 1e3:+.1f} us per residual GEMM launch)")
