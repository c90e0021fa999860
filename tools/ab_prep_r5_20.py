#!/usr/bin/env python
"""The driver's form (20 timed steps behind 3 warm-up steps) and 200 steps, headline mode: fp6 operand preparation as the stream
form (41) / the one-pass form (40) / by width (42), alternating on one box."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(2)]
res = {}
for rep in range(6):
    for variant in (41, 40):
        lib.vfm_debug_set_coarse_variant(variant)
        for steps in (20, 200):
            pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="auto")
            v, msps, cms, r = bench.timed_loop(lib, pipe, pairs, steps, 3)
            res.setdefault((variant, steps), []).append(v)
            del pipe
for k in sorted(res):
    v = sorted(res[k])
    print(f"variant {k[0]} ({'stream form' if k[0] == 41 else 'one pass'}), {k[1]:3d} steps: median {v[len(v) // 2]:7.1f}/s, all {[round(x) for x in v]}", flush=True)
lib.vfm_debug_set_coarse_variant(41)
