#!/usr/bin/env python
"""A/B of the token-stationary GEMM kernel for the QKV / fc1 products (vit_gemm_astat_kernel, vfm_debug_set_vit_gemm(-9, n)) against
the 128 x 128 LDS-tiled kernel: ViT-S/14 forward per batch size, identical bits required.  Output: profiles/r04_ab_vit_astat.txt."""
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
NMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 192
if len(sys.argv) > 2:
    lib.vfm_debug_set_vit_gemm(-10, int(sys.argv[2]))   # waves per workgroup of the token-stationary kernel
all_imgs = torch.from_numpy(rng.integers(1, 255, (NMAX, 1200, 1600, 3), dtype=np.uint8)).cuda()
tok = 337


def run(model, imgs, reps=10):
    out = model.forward(imgs).clone()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.forward(imgs)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return out, sorted(ts)[len(ts) // 2]


import os
for nimg in [n for n in [int(x) for x in os.environ.get("VFM_AB_IMAGES", "48,88,96,144,192").split(",")] if n <= NMAX]:
    imgs = all_imgs[:nimg]
    model = V.ViTS14(V.random_weights(0), 1200, 1600)
    flop = nimg * 12 * (tok * 2 * 384 * (1152 + 384 + 2 * 1536) + 6 * 2 * 2 * tok * tok * 64) + nimg * 336 * 2 * 588 * 384
    row = []
    ref = None
    for astat in (0, 1, 0, 1):
        lib.vfm_debug_set_vit_gemm(-9, 1 if astat else -1)
        out, ms = run(model, imgs)
        if ref is None:
            ref = out
        else:
            assert torch.equal(out, ref), ("the token-stationary kernel changed the features", nimg, float((out - ref).abs().max()))
        row.append(f"{'token-stationary' if astat else '128 x 128 tiles   '} {ms:6.3f} ms = {flop / ms / 1e9:5.0f} TFLOP/s")
    tp = (16 * model.patch_w + 1 + 31) // 32 * 32      # padded tokens per image
    print(f"{nimg:3d} images ({(nimg * tp // 32 + 3) // 4} groups of 128 token rows): " + " | ".join(row), flush=True)
    del model
lib.vfm_debug_set_vit_gemm(-9, 0)   # back to the default policy
