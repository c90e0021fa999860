#!/usr/bin/env python
"""A batch of N images as ONE forward against two half-batches on two streams at once (the tail of one stream's kernel under the
other's next kernel): total time and identity of the outputs."""
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
w = V.random_weights(0)
V.ViTS14.SPLIT_FROM = 0   # (this tool splits by hand; the class does it from 64 images on since)
models = [V.ViTS14(w, 1200, 1600) for _ in range(4)]   # (a workspace per stream: a model instance keeps one per batch size)
streams = [torch.cuda.Stream() for _ in range(4)]
for nimg in [int(x) for x in (sys.argv[1:] or ["24", "48", "90", "96"])]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    outs = {}

    def whole():
        return models[0].forward(imgs)

    def split(k):
        parts = torch.chunk(imgs, k, dim=0)
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        res = []
        for i, p in enumerate(parts):
            streams[i].wait_event(ev)
            with torch.cuda.stream(streams[i]):
                res.append(models[i].forward(p.contiguous()))
        for i in range(len(parts)):
            main.wait_stream(streams[i])
        return torch.cat(res, 0)

    row = []
    for name, fn in (("one forward", whole), ("two halves on two streams", lambda: split(2)), ("three thirds on three streams", lambda: split(3)),
                     ("one forward", whole), ("two halves on two streams", lambda: split(2))):
        for _ in range(3):
            o = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = fn()
        torch.cuda.synchronize()
        row.append((name, (time.perf_counter() - t0) / 10 * 1e3))
        outs[name] = o.clone()
    same = all(torch.equal(outs["one forward"], v) for v in outs.values())
    print(f"{nimg} images: " + ", ".join(f"{n}: {t:.3f} ms" for n, t in row) + f"; identical outputs: {same}", flush=True)
