"""fp6 operand preparation at C2, rows in registers (prep_chunk_kernel, variant 40) against the two-pass form that fits beside a coarse
workgroup (prep_stream_kernel, variant 41): alone on the GPU, and the bench's pipeline (20 / 200 steps) with each."""
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
q, b = pairs[0]["q_desc"], pairs[0]["b_desc"]
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device=dev)
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device=dev)
for rep in range(2):
    for variant in (40, 41):
        lib.vfm_debug_set_coarse_variant(variant)
        for flags in (24, 8):
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            print(f"variant {variant} flags {flags}: {sorted(ts)[2]:.3f} ms alone", flush=True)
for rep in range(2):
    for variant in (40, 41):
        lib.vfm_debug_set_coarse_variant(variant)
        for mode in ("mx6-half", "mx6"):
            for steps in (20, 200):
                pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
                v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3, settle=0)
                print(f"variant {variant} {mode:9s} kind {pipe._records()} steps {steps:3d}: {v:7.1f}/s  coarse kernel {cms:.3f} ms  corr {int(res['count'].item())}", flush=True)
                del pipe
lib.vfm_debug_set_coarse_variant(41)
