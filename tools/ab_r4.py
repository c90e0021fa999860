"""Round 4 A/B of the bench pipeline at C2 on D.2 data: the fp6 half-width pass with the survivor-only epilogue (record kind 8)
against the record form (kind 7), the int8 half-width pass and the full-width fp6 pass; 20 and 200 timed steps each."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(4)]
def build(coarse, fused):
    return RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=coarse, half_fused=fused)
import os
modes = (("mx6-half", None), ("mx6-half", False), ("mx6", None), ("auto", None))
for rep in range(2):
    for mode, fused in modes + ((("mx6-half", "t4"),) if True else ()):
        if fused == "t4":
            lib.vfm_debug_set_coarse_variant(31)
            fused = None
            tag = " (two chunks per barrier)"
        else:
            lib.vfm_debug_set_coarse_variant(30)
            tag = ""
        for steps in (20, 200):
            pipe = build(mode, fused)
            v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3, settle=6 if mode == "auto" else 0)
            print(f"{mode:10s}{tag} fused {str(fused):5s} kind {pipe._records()} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms  corr {int(res['count'].item())}", flush=True)
            del pipe
