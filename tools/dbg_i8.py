#!/usr/bin/env python
"""Debug / design check of the int8 image: read it back, compare with a torch emulation of the quantisation, verify the
per-(query, chunk) bound on every pair, and run the search."""
import ctypes as C
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, ops, synth  # noqa: E402

lib = _lib.load()
n, m, d = 2048, 8192, 384
p = synth.make_pair_device(n, m, d, seed=42)
Q, B = ops.PreparedRows(p["q_desc"]), ops.PreparedRows(p["b_desc"])
torch.cuda.synchronize()


def parse(buf, rows):
    rp = (rows + 255) // 256 * 256
    al = lambda v: (v + 255) // 256 * 256
    o = 0
    inv = buf[o:o + rp * 4].view(torch.float32); o += al(rp * 4)
    o += al(rp // 32 * (d // 16) * 64 * 16)
    err = buf[o:o + rp * 4].view(torch.float32); o += al(rp * 4)
    emax = buf[o:o + 4].view(torch.float32); o += 256
    gstep = buf[o:o + rp // 128 * 4].view(torch.float32); o += al(rp // 128 * 4)
    gerr = buf[o:o + rp // 128 * 4].view(torch.float32); o += al(rp // 128 * 4)
    t8 = buf[o:o + rp // 32 * (d // 32) * 64 * 16].view(torch.int8)
    t8 = t8.view(rp // 32, d // 32, 2, 32, 16).permute(0, 3, 1, 2, 4).reshape(rp, d)  # [tile][s][h][p][16] -> rows x d
    return inv, err, emax, gstep, gerr, t8


out = {}
for name, P, x in (("Q", Q, p["q_desc"]), ("B", B, p["b_desc"])):
    inv, err, emax, gstep, gerr, t8 = parse(P.buf, x.shape[0])
    r = x.shape[0]
    v = x * inv[:r, None]
    vp = torch.zeros((gstep.numel() * 128, d), device="cuda")
    vp[:r] = v
    amax = vp.view(-1, 128, d).abs().amax(dim=(1, 2))
    step = amax / 127.0
    q = torch.clamp(torch.round(vp / step.repeat_interleave(128)[:, None]), -127, 127)
    e = (vp - gstep.repeat_interleave(128)[:, None] * q).double().norm(dim=1)
    print(name, "emax", float(emax), "step equal", bool((step == gstep).all()), "tiles equal", bool((t8.float() == q).all()),
          "err vs emulated max rel", float(((err - e).abs() / e.clamp_min(1e-9))[:r].max()), "err >= emulated", bool((err[:r] >= e[:r]).all()),
          "E median", float(err[:r].median()), "gerr ok", bool((gerr == err.view(-1, 128).amax(dim=1)).all()))
    bad = (step != gstep).nonzero().flatten()
    if bad.numel():
        g = int(bad[0])
        print("   groups with a different step:", bad.numel(), "first", g, "emulated", float(step[g]), "kernel", float(gstep[g]),
              "ratio", float(step[g] / gstep[g]), "row tiles differing", int((t8.float() != q).any(dim=1).sum()))
    out[name] = (v.double(), t8[:r].double(), err[:r].double(), gstep.repeat_interleave(128)[:r].double(), gerr.repeat_interleave(128)[:r].double())
vq, q8, eq, sq, _ = out["Q"]
vb, b8, eb, sb, gb = out["B"]
S = q8 @ b8.T
t = vq @ vb.T
dev = (t - sq[:, None] * sb[None, :] * S).abs()
bound = (1.0 + 2 ** -13 + eq[:, None]) * gb[None, :] + (1 + 2 ** -13) * eq[:, None]
print("max |t - s_q s_b S|", float(dev.max()), "min slack (bound - dev)", float((bound - dev).min()), "typical bound", float(bound.mean()))
lower = (sq[:, None] * sb[None, :] * S - bound).max(dim=1, keepdim=True).values
cnt = ((sq[:, None] * sb[None, :] * S + bound) >= lower).sum(dim=1).double()
print("rows inside the window / query at m =", m, ":", float(cnt.mean()))
lib.vfm_debug_set_coarse_variant(9)
lib.vfm_debug_set_match_stats(1)
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
idx, sim = ops.match_search(Q, B, ws=ws)
torch.cuda.synchronize()
stats = (C.c_int32 * 64)()
_lib.check(lib.vfm_debug_match_stats(ws.data_ptr(), n, m, stats))
st = list(stats)
print("search: fallbacks", st[0], "refined", st[1], "cand/query", st[2] / n)
print("argmax equal:", bool((idx == t.argmax(dim=1)).all()))
