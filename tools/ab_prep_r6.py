#!/usr/bin/env python
"""Round 6: the operand preparation as prep_stream_kernel (41: every row twice), prep_chunk_kernel (40) or prep_once_kernel (43: one
read, a tile's fp16 copy in registers) inside the pipeline -- the driver's 20-step form and 200 steps, headline mode and the others."""
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
modes = sys.argv[1:] or ["mx6-half", "mx6-fused", "mx6@lifted"]
lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
for rep in range(2):
    for mode in modes:
        data = lifted if mode.endswith("@lifted") else pairs
        for steps in (20, 200):
            for variant, name in ((41, "stream form (rows twice)"), (43, "one read, fp16 copy in registers"), (40, "rows in registers, 8 waves")):
                lib.vfm_debug_set_coarse_variant(variant)
                pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2,
                                            coarse=mode.split("@")[0])
                v, msps, cms, res = bench.timed_loop(lib, pipe, data, steps, 5)
                print(f"{mode:12s} {steps:3d} steps  {name:36s}: {v:7.1f}/s  coarse kernel in the pipeline {cms:.3f} ms", flush=True)
                del pipe
lib.vfm_debug_set_coarse_variant(41)
