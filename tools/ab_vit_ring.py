#!/usr/bin/env python
"""ViT forward at N images with the LDS-tiled GEMM's ring at KB k-steps per stage x NS stages (vfm_debug_set_vit_gemm(-6, 10 KB + NS)):
23 = 48 KiB, three workgroups per compute unit (default) ... 25 = 80 KiB, two per unit, four stages ahead."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
model = V.ViTS14(V.random_weights(0), 1200, 1600)
shapes = [23, 24, 25, 26, 43]
for nimg in [int(x) for x in (sys.argv[1:] or ["36", "48", "90", "96"])]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    res = {k: [] for k in shapes}
    ref = None
    same = True
    for rep in range(3):
        for sh in shapes:
            lib.vfm_debug_set_vit_gemm(-6, sh)
            for _ in range(3):
                out = model.forward(imgs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                out = model.forward(imgs)
            torch.cuda.synchronize()
            res[sh].append((time.perf_counter() - t0) / 10 * 1e3)
            if ref is None:
                ref = out.clone()
            same = same and torch.equal(ref, out)
    print(f"{nimg} images: " + ", ".join(f"shape {sh}: {min(res[sh]):.3f} ms" for sh in shapes) + f"; identical outputs: {same}", flush=True)
lib.vfm_debug_set_vit_gemm(-6, 23)
