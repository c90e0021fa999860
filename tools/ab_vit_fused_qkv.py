#!/usr/bin/env python
"""`vit_fused_qkv` (round 6): QKV GEMM + attention kernel against vit_qkv_attention_kernel (one workgroup per (image, head)), forwards back
to back, alternating, per batch size.  python tools/ab_vit_fused_qkv.py [nimg ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "vfm-registration_amd"))
import numpy as np, torch
from vfmreg import _lib, vit as V
rng = np.random.default_rng(0)
model = V.ViTS14(V.random_weights(0), 1200, 1600)
sizes = [int(a) for a in sys.argv[1:]] or [6, 42, 48, 84, 85, 90, 96, 126]
for nimg in sizes:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    acc = {0: [], 1: []}
    for rep in range(4):
        for fused in (0, 1):
            with _lib.using(_lib.Config().set("vit_fused_qkv", 1 if fused else -1)):
                for _ in range(3): model.forward(imgs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): model.forward(imgs)
                torch.cuda.synchronize()
                acc[fused].append((time.perf_counter() - t0) / 10 * 1e3)
    a, b = sorted(acc[0])[1], sorted(acc[1])[1]
    print(f"{nimg:4d} images: two kernels {a:.3f} ms, one {b:.3f} ms ({(b / a - 1) * 100:+.1f} %)   per image {a / nimg * 1e3:.1f} / {b / nimg * 1e3:.1f} us", flush=True)
