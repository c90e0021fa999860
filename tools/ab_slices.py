"""Interleaved A/B of the pipelined registration throughput for forced slice counts of the coarse pass (0 = heuristic)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + j) for j in range(2)]
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
cfgs = [int(a) for a in sys.argv[1:]] or [0, 55]
pipe = RegistrationPipeline(n, m, d, overlap_ransac=True)
res = {c: [] for c in cfgs}
for rnd in range(6):
    for c in cfgs:
        lib.vfm_debug_set_coarse_slices(c)
        steps = 24
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            p = pairs[i % 2]
            out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ready)
        pipe.synchronize()
        torch.cuda.synchronize()
        if rnd:
            res[c].append(steps / (time.perf_counter() - t0))
lib.vfm_debug_set_coarse_slices(0)
for c in cfgs:
    r = sorted(res[c])
    print(f"slices {c:3d}: median {r[len(r)//2]:.1f} reg/s  (min {r[0]:.1f} max {r[-1]:.1f})")
