"""With the side streams shared per process: number of solve streams (2 / 3), under the default 4 hardware queues and under
GPU_MAX_HW_QUEUES=8 (set by the caller), auto mode (fp6 half-width on D.2 data) and int8 full width."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/vfm-registration_amd")
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(2)]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for rep in range(2):
    for mode in ("auto", "int8"):
        for ns in (2, 3):
            pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=ns, coarse=mode)
            v20 = bench.timed_loop(lib, pipe, pairs, 20, 3, settle=4 if mode == "auto" else 0)[0]
            v200, _, cms, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
            print(f"{mode:6s} solve streams {ns}: 20 steps {v20:7.1f}/s, 200 steps {v200:7.1f}/s, coarse {cms:.3f} ({bench.pass_name(pipe)[:28]})", flush=True)
            del pipe
