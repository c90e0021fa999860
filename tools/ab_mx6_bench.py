"""A/B of the pipeline with the coarse pass pinned to int8 / mx6 / int8-top2 / auto on D.2, lifted and lifted + common data (C2 size)."""
import sys, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(2)]
lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
lifted0 = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1) for p in range(2)]
def build(coarse):
    return RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=coarse)
for name, data in (("D.2", pairs), ("lifted+common", lifted), ("lifted", lifted0)):
    for mode in ("int8", "mx6", "int8-top2", "mx6-top2", "auto", "int8-half", "mx6-half"):
        for steps in (20, 200):
            pipe = build(mode)
            v, msps, cms, res = bench.timed_loop(lib, pipe, data, steps, 3, settle=6 if mode == "auto" else 0)
            print(f"{name:14s} {mode:10s} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms  pass {bench.pass_name(pipe)[:40]}  corr {int(res['count'].item())}", flush=True)
            del pipe
