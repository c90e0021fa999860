#!/usr/bin/env python
"""Design study (torch only): candidates per query of an int8 coarse pass vs the clipping point, at C2's map size."""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import synth  # noqa: E402

n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
q = p["q_desc"][:2048]
b = p["b_desc"]
t = q.double() @ b.double().T
tmax = t.max(dim=1, keepdim=True).values
for w in (2.5e-3, 0.02, 0.03, 0.04, 0.05):
    print(f"exact scores: rows within {w} of the maximum: {float((t >= tmax - w).sum(dim=1).double().mean()):.1f} / query")
for c in (3.9, 4.5, 5.0, 5.5, 6.0):
    s = c / (127.0 * math.sqrt(d))
    def quant(x):
        k = torch.clamp(torch.round(x / s), -127, 127)
        return k, (x - s * k).double().norm(dim=1)
    qa, ea = quant(q)
    qb, eb = quant(b)
    for cut_mult in (1.15, 1.3):
        ecut = cut_mult * float(eb.median())
        dirty_b = int((eb > ecut).sum())
        dirty_q = float((ea > ecut).double().mean())
        S = qa.double() @ qb.double().T
        S[:, eb > ecut] = -1e18
        W = 2 * (ea + ecut) * 1.001 / (s * s)
        cnt = (S >= S.max(dim=1, keepdim=True).values - W[:, None]).sum(dim=1).double()
        clean = ea <= ecut
        print(f"clip {c}: E median {float(eb.median()):.5f}, cut {ecut:.5f}: dirty map rows {dirty_b}, dirty queries {100 * dirty_q:.2f} %, "
              f"window {float(W[clean].mean() * s * s):.4f}, candidates/query (clean queries) {float(cnt[clean].mean()):.1f}, "
              f"p99 {float(cnt[clean].quantile(0.99)):.0f}", flush=True)
