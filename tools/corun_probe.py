#!/usr/bin/env python
"""Does a THIN memory-bound kernel run beside the coarse kernels?  Stream A: 20 coarse launches back to back (record kind argv[1]);
stream B: torch element-wise kernels (256-thread workgroups, few registers) that move about what an operand preparation moves
(read 2 x 338 MB, write 170 MB per round).  Times alone and together."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT / "tools"))
from vfmreg import _lib, synth  # noqa: E402
import dev_mx6 as D  # noqa: E402

lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
q, b = p["q_desc"], p["b_desc"]
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
x = torch.randn(340 * 1024 * 1024 // 4, device="cuda")
y = torch.empty(170 * 1024 * 1024 // 2, device="cuda", dtype=torch.float16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def coarse_run(rec, qb, bb, ws, k):
    for _ in range(k):
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), rec, gate, sa.cuda_stream))


def thin_round():
    s = x.sum()                              # read 338 MB
    y.copy_(x[: y.numel()])                  # read 170 MB, write 170 MB (fp32 -> fp16)
    return s


for rec in (3, 7, 0, 5):
    qb, bb = D.prepare(b, q, 8 if rec in (5, 7) else 0)
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    res = {}
    for what in ("coarse alone", "thin alone", "together"):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        if what != "thin alone":
            e[0].record(sa)
            with torch.cuda.stream(sa):
                coarse_run(rec, qb, bb, ws, 20)
            e[1].record(sa)
        if what != "coarse alone":
            e[2].record(sb)
            with torch.cuda.stream(sb):
                for _ in range(40):
                    thin_round()
            e[3].record(sb)
        torch.cuda.synchronize()
        res[what] = (e[0].elapsed_time(e[1]) / 20 if what != "thin alone" else None, e[2].elapsed_time(e[3]) / 40 if what != "coarse alone" else None)
    print(f"records {rec}: coarse alone {res['coarse alone'][0]:.3f} ms; thin round alone {res['thin alone'][1]:.3f} ms; together: coarse {res['together'][0]:.3f} ms, "
          f"thin round {res['together'][1]:.3f} ms")
