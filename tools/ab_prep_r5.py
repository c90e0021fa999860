#!/usr/bin/env python
"""Round 5, with three query tiles per wave in the coarse kernel (222 registers x 2 waves per SIMD: nothing fits beside it any more):
the operand preparation as prep_stream_kernel (fits beside round 4's coarse workgroup; default) or prep_chunk_kernel (rows in
registers, one pass; interleaved / persistent launch), on a stream of its own (default) or on the coarse stream."""
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
modes = sys.argv[1:] or ["mx6-half"]
lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
for rep in range(2):
    for mode in modes:
        data = lifted if mode.endswith("@lifted") else pairs
        for variant, sched, name in ((41, None, "stream form"), (40, 2, "rows in registers, one workgroup per group")):
            for op in ((True, False) if mode == "mx6-half" else (True,)):
                lib.vfm_debug_set_coarse_variant(variant)
                pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=op, solve_streams=2,
                                            coarse=mode.split("@")[0], prep_schedule=sched)
                v, msps, cms, res = bench.timed_loop(lib, pipe, data, 200, 5)
                print(f"{mode:14s} {name:45s} preparation on its own stream {int(op)}: {v:7.1f}/s  coarse kernel {cms:.3f} ms", flush=True)
                del pipe
lib.vfm_debug_set_coarse_variant(41)
