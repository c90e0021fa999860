#!/usr/bin/env python
"""ViTS14.SPLIT_FROM (opt-in): one forward against two half-batches on two side streams (helper thread), single synchronised forwards
and forwards back to back."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "vfm-registration_amd"))
import numpy as np, torch
from vfmreg import vit as V
rng = np.random.default_rng(0)
model = V.ViTS14(V.random_weights(0), 1200, 1600)
for nimg in (48, 66, 72, 90, 96, 120):
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    row = []
    for split in (0, 64, 0, 64):
        V.ViTS14.SPLIT_FROM = split
        for _ in range(3): model.forward(imgs)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); model.forward(imgs); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(10): model.forward(imgs)
        torch.cuda.synchronize()
        row.append((split, sorted(ts)[5] * 1e3, (time.perf_counter() - t0) / 10 * 1e3))
    print(nimg, "images:", ", ".join(f"split from {s}: single {a:.3f} ms, back to back {b:.3f} ms" for s, a, b in row), flush=True)
V.ViTS14.SPLIT_FROM = 0
