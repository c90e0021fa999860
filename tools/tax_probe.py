#!/usr/bin/env python
"""What each side stage of the bench pipeline costs the steady state: C2 registrations per second with stages replaced by
no-ops (results are garbage in those arms -- this only apportions time).  300 steps per arm after 30 warm-up steps."""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402
from vfmreg import synth  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + i, device=dev) for i in range(2)]
real = {k: getattr(lib, k) for k in ("vfm_ransac_corr", "vfm_match_search_finish_gated_r", "vfm_threshold_compact",
                                      "vfm_match_prepare2_gated", "vfm_match_search_rescans_async",
                                      "vfm_match_search_coarse_gated_r")}
noop = lambda *a: 0  # noqa: E731


def run(label, off, steps=150, warm=12):
    for k, f in real.items():
        setattr(lib, k, f)
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2)
    for i in range(warm):   # every buffer set gets real contents first
        p = pairs[i % 2]
        pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        if i == 5:
            for k in off:
                setattr(lib, k, noop)
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        p = pairs[i % 2]
        pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    for k, f in real.items():
        setattr(lib, k, f)
    del pipe
    return ms


S = ["vfm_match_search_finish_gated_r", "vfm_match_search_rescans_async"]
T = ["vfm_threshold_compact"]
R = ["vfm_ransac_corr"]
P = ["vfm_match_prepare2_gated"]
ARMS = [("full", []), ("no ransac", R), ("no finish (select/rescan/refine/rescore)", S), ("no prepare", P),
        ("no finish, no threshold, no ransac", S + T + R), ("coarse only", S + T + R + P),
        ("no coarse (side stages only)", ["vfm_match_search_coarse_gated_r"])]
res = {a: [] for a, _ in ARMS}
for rnd in range(5):   # arms interleaved: the box's clock drifts over seconds
    for a, off in ARMS:
        res[a].append(run(a, off))
for a, v in res.items():
    v.sort()
    print(json.dumps({"arm": a, "ms_per_step_median": round(v[len(v) // 2], 4), "min": round(v[0], 4), "max": round(v[-1], 4)}),
          flush=True)
