#!/usr/bin/env python
"""Round 6 A/B of the headline coarse kernel (record kind 8): vfm_config "mx6_tune" bit 0 = s_setprio 1 for waves 4 - 7, bit 1 = ring of
five steps; the kernel alone (HIP events around 40 back-to-back calls) and the bench pipeline (20 / 200 steps)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import bench  # noqa: E402
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p) for p in range(2)]
p = pairs[0]
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
_lib.check(lib.vfm_match_prepare2_gated_p(p["b_desc"].data_ptr(), m, bb.data_ptr(), p["q_desc"].data_ptr(), n, qb.data_ptr(), d, 24, st))
tunes = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 3]
for rep in range(2):
    for tune in tunes:
        with _lib.using(_lib.Config(mx6_tune=tune)):
            for _ in range(5):
                _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
            e1.record()
            torch.cuda.synchronize()
            alone = e0.elapsed_time(e1) / 40
            pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="mx6-half")
            v20, _, c20, _ = bench.timed_loop(lib, pipe, pairs, 20, 3)
            v200, _, c200, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
            del pipe
        print(f"mx6_tune {tune}: coarse call alone {alone:.4f} ms | pipeline 20 steps {v20:7.1f}/s (kernel {c20:.3f}), 200 steps {v200:7.1f}/s (kernel {c200:.3f})", flush=True)
