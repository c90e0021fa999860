#!/usr/bin/env python
"""ViT-S/14 on 6 x 1200 x 1600: 87 launches issued one by one against the same launches replayed from a captured HIP graph."""
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
out = model.forward(imgs)
grids = torch.empty_like(out)
for _ in range(3):
    model.forward(imgs, out=grids)
torch.cuda.synchronize()
ref = grids.clone()


def med(fn, n=40):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


print("direct launches: %.3f ms (min %.3f)" % med(lambda: model.forward(imgs, out=grids)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model.forward(imgs, out=grids)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    model.forward(imgs, out=grids)
grids.zero_()
g.replay()
torch.cuda.synchronize()
print("graph replay:    %.3f ms (min %.3f)" % med(g.replay), " same output:", bool(torch.equal(grids, ref)))
print("direct launches: %.3f ms (min %.3f)" % med(lambda: model.forward(imgs, out=grids)))
