"""Does the pipeline's rate depend on which hardware queues its streams land on?  HIP hands new streams hardware queues round
robin (4 by default): k dummy streams created before the pipeline shift the assignment of its three streams."""
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
created = 0
keep = []
for k in (0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1):
    for _ in range(k):
        keep.append(torch.cuda.Stream())
        created += 1
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode, private_streams=(len(sys.argv) > 2 and sys.argv[2] == "private"))
    created += 3
    keep.append(pipe.prep_stream); keep.extend(pipe.solve_streams)
    v20 = bench.timed_loop(lib, pipe, pairs, 20, 3)[0]
    v200, _, cms, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
    print(f"streams created before this pipeline's: {created - 3:3d} (mod 4 = {(created - 3) % 4}): 20 steps {v20:7.1f}/s, 200 steps {v200:7.1f}/s, coarse {cms:.3f}", flush=True)
    del pipe
