"""Full-width fp6 pass with and without the pilot rescan (VFM_RECORDS_MX6_PILOT) in the bench's pipeline: lifted descriptors (with and
without the common component), D.2; 20 / 200 steps."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
sets = (("lifted + common", [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(4)]),
        ("lifted", [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1) for p in range(4)]),
        ("D.2", [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(4)]))
for rep in range(2):
    for name, pairs in sets:
        for mode in ("mx6", "mx6-pilot"):
            for steps in (20, 200):
                pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
                v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3, settle=0)
                pipe._poll_feedback()
                print(f"{name:16s} {mode:10s} kind {pipe._records()} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms  rescans/query {(pipe.last_rescans or 0) / n:.1f}  corr {int(res['count'].item())}", flush=True)
                del pipe
