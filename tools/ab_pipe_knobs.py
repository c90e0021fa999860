#!/usr/bin/env python
"""200-step C2 pipelines on one box: the solve stage on 1 / 2 / 3 side streams, for the headline mode and lifted descriptors."""
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
lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
for rep in range(2):
    for name, data in (("D.2", pairs), ("lifted", lifted)):
        for ss in ([int(x) for x in sys.argv[1:]] or [1, 2, 3, 4]):
            pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=ss, coarse="auto")
            v, msps, cms, res = bench.timed_loop(lib, pipe, data, 200, 5)
            print(f"{name:7s} solve streams {ss}: {v:7.1f}/s  coarse kernel {cms:.3f} ms", flush=True)
            del pipe
