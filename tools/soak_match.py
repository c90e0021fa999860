#!/usr/bin/env python
"""Randomised soak of the FAST inner-product search against the oracle at sizes where every mechanism of the sparse path is
active at once (seed units: >= 256 map chunks; ragged n / m; LDS record buffer overflow; duplicate-rich rows; zero rows).
Every trial runs three calls: ungated (fp16 pass), gated with gate = -inf (int8 pass for d = 256 / 384, every query
resolved) and gated at 0.8 (unresolved rows must score below 0.8 in the oracle, the others must equal it)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402
from vfmreg import ops  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
bad = 0
for t in range(trials):
    d = int(rng.choice([128, 256, 384]))
    n = int(rng.integers(513, 6000))
    m = int(rng.integers(32768, 150000))
    b = rng.standard_normal((m, d)).astype(np.float32)
    q = rng.standard_normal((n, d)).astype(np.float32)
    kind = t % 4
    if kind >= 1:   # matched queries
        pick = rng.integers(0, m, n)
        q = b[pick] + np.float32(rng.choice([0.05, 0.3])) * rng.standard_normal((n, d)).astype(np.float32)
    if kind >= 2:   # clusters of near-duplicates / exact duplicates spread over the map
        for _ in range(int(rng.integers(1, 6))):
            base = rng.standard_normal(d).astype(np.float32)
            k = int(rng.integers(2, 400))
            rows = rng.choice(m, k, replace=False)
            b[rows] = base + np.float32(rng.choice([0.0, 1e-4, 5e-3, 3e-2])) * rng.standard_normal((k, d)).astype(np.float32)
            hit = rng.choice(n, min(n, int(rng.integers(1, 300))), replace=False)
            q[hit] = base + 1e-3 * rng.standard_normal((len(hit), d)).astype(np.float32)
    if kind == 3:   # zero rows on both sides, a low-dimensional (smooth) block
        b[rng.choice(m, 500, replace=False)] = 0
        q[rng.choice(n, n // 10, replace=False)] = 0
        A = rng.standard_normal((8, d)).astype(np.float32)
        rows = rng.choice(m, 3000, replace=False)
        b[rows] = rng.random((3000, 8)).astype(np.float32) @ A
        hit = rng.choice(n, n // 4, replace=False)
        q[hit] = rng.random((len(hit), 8)).astype(np.float32) @ A
    qd, bd = torch.from_numpy(q).cuda(), torch.from_numpy(b).cuda()
    qn, _ = orc.l2norm_rows(q)
    bn, _ = orc.l2norm_rows(b)
    ir, sr = orc.match_ip_top1(qn, bn)
    ok, ms = True, []
    for gate in (None, float("-inf"), 0.8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx, sim = ops.match_ip_top1(qd, bd, ops.FAST, gate=gate)
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        gi, gs = idx.cpu().numpy(), sim.cpu().numpy()
        solved = gi >= 0
        ok &= np.array_equal(gi[solved], ir[solved]) and np.array_equal(gs[solved], sr[solved])
        ok &= bool(solved.all()) if gate != 0.8 else bool((sr[~solved] < 0.8).all())
    bad += not ok
    print(f"trial {t:2d} kind {kind} n={n:5d} m={m:6d} d={d}: {'ok' if ok else 'MISMATCH'}  gpu ungated / gate -inf / gate 0.8: "
          f"{ms[0]:7.1f} {ms[1]:7.1f} {ms[2]:7.1f} ms", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
