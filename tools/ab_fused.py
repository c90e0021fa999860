"""A/B under the stable pipeline: the half-width pass with its selection fused into the coarse kernel (VFM_RECORDS_HALF_FUSED) against
records + selection kernel, int8 image, bench's pipeline."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(2)]
for rep in range(3):
    for fused in (False, True):
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="int8-half", half_fused=fused)
        v20 = bench.timed_loop(lib, pipe, pairs, 20, 3)[0]
        v200, _, cms, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
        print(f"int8-half fused {int(fused)}: 20 steps {v20:7.1f}/s, 200 steps {v200:7.1f}/s, coarse {cms:.3f}", flush=True)
        del pipe
