#!/usr/bin/env python
"""Randomised soak of row A6 (vfm_match_mutual_pairs / vfm_match_mutual_l2) against the oracle's find_correspondences / nn_l2:
random shapes and widths (the int8 path for d = 256 ... 768, the fp16 path elsewhere), un-normalised rows of mixed norms,
planted mutual pairs, duplicates, zero rows.   python tools/soak_l2.py [trials] [seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402
from vfmreg import ops  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for t in range(trials):
    d = int(rng.choice([33, 128, 200, 256, 384, 384, 512, 768]))
    n = int(rng.integers(1, 5000))
    m = int(rng.integers(1, 30000))
    kind = t % 4
    a = (rng.standard_normal((n, d)) * rng.uniform(0.1, 4.0, (n, 1))).astype(np.float32)
    b = (rng.standard_normal((m, d)) * rng.uniform(0.1, 4.0, (m, 1))).astype(np.float32)
    if kind >= 1 and min(n, m) > 4:       # planted mutual pairs
        k = min(n, m) // 2
        a[:k] = b[rng.permutation(m)[:k]] + float(rng.choice([0.01, 0.05, 0.3])) * rng.standard_normal((k, d)).astype(np.float32)
    if kind == 2 and m > 50:              # duplicates and a row that sets the common scale
        b[m // 2:] = b[rng.integers(0, m // 2, m - m // 2)]
        b[3] *= 50.0
    if kind == 3:                         # zero rows, unit rows
        b[::17] = 0.0
        a[::23] = 0.0
        a /= np.maximum(np.linalg.norm(a, axis=1, keepdims=True), 1e-20)
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    i0_ref, i1_ref = orc.find_correspondences(a, b, mutual_filter=True)
    nn_ref, dist_ref = orc.nn_l2(a, b)
    nn_ba_ref, _ = orc.nn_l2(b, a)
    i0, i1, cnt, nn_ab, d2 = ops.match_mutual_pairs(ad, bd, want_nn=True)
    nn2_ab, _, nn2_ba = ops.match_mutual_l2(ad, bd)
    torch.cuda.synchronize()
    k = int(cnt.item())
    ok = (k == len(i0_ref) and np.array_equal(i0[:k].cpu().numpy(), i0_ref) and np.array_equal(i1[:k].cpu().numpy(), i1_ref)
          and np.array_equal(nn_ab.cpu().numpy(), nn_ref) and np.array_equal(np.sqrt(d2.cpu().numpy()), dist_ref)
          and np.array_equal(nn2_ab.cpu().numpy(), nn_ref) and np.array_equal(nn2_ba.cpu().numpy(), nn_ba_ref))
    bad += 0 if ok else 1
    print(f"trial {t}: kind {kind} n {n} m {m} d {d}: {k} mutual pairs -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{trials} trials, {bad} mismatches")
