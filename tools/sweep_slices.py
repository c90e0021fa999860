"""Sweep of the coarse kernel's map slices per query block (vfm_debug_set_coarse_slices; 0 = the heuristic) for the modes argv names:
coarse kernel in the pipeline and registrations/s over 200 steps (stable pipeline: +-0.5 %)."""
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
for mode in (sys.argv[1:] or ["mx6-half"]):
    for s in [int(x) for x in __import__("os").environ.get("SLICES", "0,6,13,19,26,32,38,45,51,58,64,0").split(",")]:
        lib.vfm_debug_set_coarse_slices(s)
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
        v20 = bench.timed_loop(lib, pipe, pairs, 20, 3)[0]
        v, _, cms, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
        print(f"{mode:10s} slices {s:2d}: 20 steps {v20:7.1f}/s, 200 steps {v:7.1f}/s, coarse {cms:.3f} ms", flush=True)
        del pipe
lib.vfm_debug_set_coarse_slices(0)
