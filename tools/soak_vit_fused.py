#!/usr/bin/env python
"""Soak of the two fused ViT kernels beyond the suite's fixed shapes: random image sizes (1 .. 12 token tiles per image), batch sizes and depths;
vit_qkv_attention_kernel and vit_mlp_kernel forced on (alone and together) against the GEMM + attention kernels -- every output bit.
python tools/soak_vit_fused.py [trials] [seed]"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, vit as V  # noqa: E402
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 606)
bad = 0
for t in range(trials):
    H = int(rng.integers(20, 90)) * 14
    pw = int(rng.integers(1, 24))                      # patch-grid width: 16 pw + 1 tokens
    W = int(np.ceil((pw + 0.2) * 14 * H / 224.0))
    B = int(rng.integers(1, 20))
    depth = int(rng.integers(1, 4))
    w = V.random_weights(seed=int(rng.integers(1 << 30)), dim=384, depth=depth, mlp=1536)
    imgs = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).cuda()
    model = V.ViTS14(w, H, W, device="cuda")
    tiles = (16 * model.patch_w + 1 + 31) // 32
    with _lib.using(_lib.Config().set("vit_fused_qkv", -1).set("vit_fused_mlp", -1)):
        ref = model.forward(imgs).clone()
    ok = []
    for qkv, mlp in ((1, -1), (-1, 1), (1, 1)):
        with _lib.using(_lib.Config().set("vit_fused_qkv", qkv).set("vit_fused_mlp", mlp)):
            out = model.forward(imgs).clone()
        torch.cuda.synchronize()
        ok.append(bool(torch.equal(ref, out)) and bool(torch.isfinite(out).all()))
    bad += not all(ok)
    print(f"trial {t}: {B} images {H} x {W} ({tiles} token tiles{', the two kernels' if tiles > 12 else ''}), depth {depth}: qkv / mlp / both {ok} -> {'ok' if all(ok) else 'MISMATCH'}", flush=True)
    del model
print(f"{trials} trials, {bad} mismatches")
