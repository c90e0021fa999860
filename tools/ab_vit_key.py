#!/usr/bin/env python
"""A ViT config key at two values, forwards back to back, alternating, per batch size: python tools/ab_vit_key.py KEY A B [nimg ...]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "vfm-registration_amd"))
import numpy as np, torch
from vfmreg import _lib, vit as V
key, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
model = V.ViTS14(V.random_weights(0), 1200, 1600)
for nimg in [int(a) for a in sys.argv[4:]] or [6, 42, 84, 90]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    acc = {va: [], vb: []}
    outs = {}
    for rep in range(4):
        for v in (va, vb):
            with _lib.using(_lib.Config().set(key, v)):
                for _ in range(3): outs[v] = model.forward(imgs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10): model.forward(imgs)
                torch.cuda.synchronize()
                acc[v].append((time.perf_counter() - t0) / 10 * 1e3)
    a, b = sorted(acc[va])[1], sorted(acc[vb])[1]
    print(f"{nimg:4d} images: {key} = {va}: {a:.3f} ms, = {vb}: {b:.3f} ms ({(b / a - 1) * 100:+.1f} %)   same bits: {bool(torch.equal(outs[va], outs[vb]))}", flush=True)
