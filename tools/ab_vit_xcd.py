#!/usr/bin/env python
"""A/B of the XCD-consistent tile mapping of the ViT kernels (vfm_debug_set_vit_gemm(-3 / -4)): ViT-S/14 on 6 x 1200 x 1600,
interleaved arms."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.integers(1, 255, (6, 1200, 1600, 3), dtype=np.uint8)).cuda()
model = V.ViTS14(V.random_weights(0), 1200, 1600)
ref = None
arms = [("round 2: plain mapping", (-4,)), ("tile mt on XCD mt % 8 (default)", (-3,))]
for name, modes in arms + arms:
    for m in modes:
        lib.vfm_debug_set_vit_gemm(m, 0)
    out = model.forward(imgs)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.forward(imgs)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"{name:45s}: {ts[len(ts) // 2]:.3f} ms (min {ts[0]:.3f})   max|diff vs round 2| {float((out - ref).abs().max()):.2e}", flush=True)
lib.vfm_debug_set_vit_gemm(-3, 0)
