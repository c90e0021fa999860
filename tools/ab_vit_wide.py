#!/usr/bin/env python
"""ViT forward at N images: the residual GEMMs (proj, fc2) of the LDS-tiled path as 128 x 128 tiles (vfm_debug_set_vit_gemm(-17, 0))
against one 128 x 384 tile per workgroup (1), alternating on the same box."""
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
for nimg in [int(x) for x in (sys.argv[1:] or ["36", "48", "90", "96"])]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    res = {0: [], 1: []}
    outs = {}
    for rep in range(3):
        for wide in (0, 1):
            lib.vfm_debug_set_vit_gemm(-17, wide)
            for _ in range(3):
                out = model.forward(imgs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                out = model.forward(imgs)
            torch.cuda.synchronize()
            res[wide].append((time.perf_counter() - t0) / 10 * 1e3)
            outs[wide] = out.clone()
    print(f"{nimg} images: 128 x 128 tiles {min(res[0]):.3f} ms, 128 x 384 tiles {min(res[1]):.3f} ms, identical outputs: {torch.equal(outs[0], outs[1])}", flush=True)
lib.vfm_debug_set_vit_gemm(-17, 0)
