"""Latency of one registration in the real-data regime (queries = 5 m voxel subset: a few hundred to ~2k rows)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import synth, _lib
import os
_lib.load().vfm_debug_set_coarse_variant(int(os.environ.get('VFM_VARIANT', '0')))
if os.environ.get('VFM_I8_MIN'):
    _lib.load().vfm_debug_set_i8_min_queries(int(os.environ['VFM_I8_MIN']))
from vfmreg.pipeline import RegistrationPipeline
for (n, m) in ((300, 50000), (1500, 100000), (2000, 200000), (20000, 200000)):
    p = synth.make_pair_device(n, m, 384, seed=1)
    pipe = RegistrationPipeline(n, m, 384, n_iter=50000)
    for mode in ("fresh map", "reuse_map"):
        ts = []
        for r in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], reuse_map=(mode == "reuse_map"))
            b.record(); b.synchronize()
            if r > 1: ts.append(a.elapsed_time(b))
        ts.sort()
        print(f"n={n:6d} m={m:7d} {mode:10s}: {ts[len(ts)//2]*1e3:8.1f} us per registration ({int(out['count'].item())} correspondences)")
