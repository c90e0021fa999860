"""Does the pipeline's rate depend on where its buffers lie?  Builds the bench's pipeline several times (the caching allocator hands
out different blocks), prints the sets' buffer addresses next to the rate."""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "int8-half"
junk = []
for rep in range(10):
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
    v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, 200, 3)
    addrs = " ".join(f"[b {r.bprep.data_ptr() >> 12 & 0xffff:04x} q {r.qprep.data_ptr() >> 12 & 0xffff:04x} w {r.sws.data_ptr() >> 12 & 0xffff:04x}]" for r in pipe.sets)
    print(f"rep {rep}: {v:7.1f}/s coarse {cms:.3f}  sets (address bits 12..27): {addrs}", flush=True)
    del pipe
    if rep % 3 == 2:
        junk.append(torch.empty((rep + 1) * 3_000_000, device=dev))   # perturb the allocator
