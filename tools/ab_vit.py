#!/usr/bin/env python
"""A/B of the ViT GEMM wave tile / prefetch depth (vfm_debug_set_vit_gemm): forward time of ViT-S/14 on 6 x 1200 x 1600."""
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
for narrow, wide in ((108, 208), (108, 108), (116, 116), (108, 208), (108, 108)):   # (108, 208) = round 1, (108, 108) = default
    lib.vfm_debug_set_vit_gemm(narrow, wide)
    out = model.forward(imgs)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.forward(imgs)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    print(f"narrow {narrow} wide {wide}: {ts[len(ts) // 2]:.3f} ms   max|diff vs first cfg| {float((out - ref).abs().max()):.2e}", flush=True)
lib.vfm_debug_set_vit_gemm(0, 0)
