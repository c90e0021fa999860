#!/usr/bin/env python
"""Randomised soak of the half-width pass (VFM_RECORDS_HALF = 3, fused = 4) against best-score records (0) of the gated family:
random shapes, widths, gates and data kinds; checks the gate contract pairwise (same answer where both resolve; whatever only
one resolves lies below the gate; the matches a caller keeps -- similarity >= gate -- are identical)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402

lib = _lib.load()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = torch.cuda.current_stream().cuda_stream
bad = 0
for t in range(trials):
    d = int(rng.choice([256, 384, 384, 512, 768]))
    n = int(rng.integers(1, 7000))
    m = int(rng.integers(1, 60000))
    gate = float(np.nextafter(np.float32(rng.choice([0.5, 0.8, 0.8, 0.95])), np.float32(-np.inf)))
    kind = rng.choice(["planted", "alike", "duplicates", "halves"])
    g = torch.Generator(device="cuda")
    g.manual_seed(int(rng.integers(1 << 30)))
    b = torch.randn((m, d), generator=g, device="cuda")
    pick = torch.randint(0, m, (n,), generator=g, device="cuda")
    q = b[pick] + float(rng.choice([0.1, 0.3, 0.6])) * torch.randn((n, d), generator=g, device="cuda")
    if kind == "alike":
        base = torch.randn((1, d), generator=g, device="cuda")
        b = base + 0.3 * b
        q = base + 0.3 * q
    elif kind == "duplicates":
        b = b[torch.randint(0, max(1, m // 50), (m,), generator=g, device="cuda")].clone()
        q = b[pick].clone()
    elif kind == "halves":
        b[: m // 2, : d // 2] *= 1e-3
        q[::2, d // 2:] *= 1e-3
    q[torch.rand(n, generator=g, device="cuda") < 0.3] = torch.randn((d,), generator=g, device="cuda")
    q, b = q.contiguous(), b.contiguous()
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    _lib.check(lib.vfm_match_prepare2_gated(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, st))
    res = {}
    for records in (0, 3, 4):
        idx = torch.empty(n, dtype=torch.int64, device="cuda")
        sim = torch.empty(n, dtype=torch.float32, device="cuda")
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, gate, st))
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records, st))
        torch.cuda.synchronize()
        res[records] = (idx, sim)
    i0, s0 = res[0]
    ok = True
    for r in (3, 4):
        i, s = res[r]
        both = (i >= 0) & (i0 >= 0)
        ok &= bool(torch.equal(i[both], i0[both]) and torch.equal(s[both], s0[both]))
        ok &= bool((s0[(i0 >= 0) & (i < 0)] < gate).all())          # what only best-score records resolve lies below the gate
        ok &= int(((i >= 0) & (i0 < 0)).sum()) == 0                 # the half-width pass never resolves more
        keep, keep0 = s >= gate, s0 >= gate
        ok &= bool(torch.equal(keep, keep0) and torch.equal(i[keep], i0[keep0]))
    bad += 0 if ok else 1
    print(f"trial {t}: d {d} n {n} m {m} gate {gate:.3f} {kind}: kept {int((s0 >= gate).sum())}, resolved 0/3/4 "
          f"{int((i0 >= 0).sum())}/{int((res[3][0] >= 0).sum())}/{int((res[4][0] >= 0).sum())} -> {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{trials} trials, {bad} mismatches")
sys.exit(1 if bad else 0)
