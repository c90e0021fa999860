"""A/B of the preparation kernel's launch shape (persistent / interleaved) in the bench's pipeline, per coarse mode."""
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
modes = sys.argv[1:] or ["int8-half", "mx6-half", "mx6", "int8"]
for rep in range(3):
    for mode in modes:
        for sched in (1, 2):
            for steps in (20, 200):
                pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode, prep_schedule=sched)
                v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3)
                print(f"{mode:10s} schedule {sched} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms", flush=True)
                del pipe
