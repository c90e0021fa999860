#!/usr/bin/env python
"""Randomised soak of the ViT's GEMM kernels against each other: for random image sizes, batch sizes, depths and widths the forward with the
token-stationary QKV / fc1 kernel forced on, with the LDS-tiled kernel forced on everywhere, and with the direct kernel only must agree
bit for bit.   python tools/soak_vit_astat.py [trials] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
lib = _lib.load()
bad = 0
for t in range(trials):
    dim = int(rng.choice([128, 256, 384, 384, 384, 512]))
    depth = int(rng.integers(1, 4))
    H = int(rng.integers(20, 90)) * 14
    W = int(H * rng.uniform(0.8, 2.0))   # (16 x 12 ... 16 x 32 patches)
    B = int(rng.integers(1, 40))
    w = V.random_weights(seed=int(rng.integers(1 << 30)), dim=dim, depth=depth, mlp=4 * dim)
    imgs = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    model = V.ViTS14(w, H, W, device="cuda")
    outs = []
    for astat, lds in ((-1, 0), (-1, 1), (1, 1), (1, 0)):
        lib.vfm_debug_set_vit_gemm(-9, astat)
        lib.vfm_debug_set_vit_gemm(-5, lds)
        outs.append(model.forward(imgs).clone())
    torch.cuda.synchronize()
    ok = all(torch.equal(outs[0], o) for o in outs[1:]) and bool(torch.isfinite(outs[0]).all())
    bad += 0 if ok else 1
    print(f"trial {t}: dim {dim} depth {depth} {B} x {H}x{W} ({model.patch_w} patch columns) -> {'ok' if ok else 'MISMATCH'}", flush=True)
    del model, imgs, outs
lib.vfm_debug_set_vit_gemm(-9, 0)
lib.vfm_debug_set_vit_gemm(-5, 256)
print(f"{trials} trials, {bad} mismatches")
