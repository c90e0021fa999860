"""The bench's pipeline with K solve streams (K + 1 buffer sets): is the chain finish stage -> RANSAC of a pair, stretched under the next pairs'
coarse kernels, what bounds the cycle?  One configuration per process (side streams are created once per process):
    [GPU_MAX_HW_QUEUES=8] python tools/ab_solve_streams.py <K>"""
import os
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
d2 = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(4)]
lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(4)]
for name, pairs, mode in (("D.2", d2, "auto"), ("lifted + common", lifted, "auto"), ("D.2 full-width fp6", d2, "mx6")):
    for steps in (20, 200):
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=K, coarse=mode)
        v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3, settle=6 if mode == "auto" else 0)
        print(f"solve streams {K} hw queues {os.environ.get('GPU_MAX_HW_QUEUES', 'default')} | {name:20s} kind {pipe._records()} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms", flush=True)
        del pipe
