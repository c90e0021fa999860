"""Map slices of the fused fp6 half-width kernel (record kind 8) at C2: coarse-kernel time alone and registrations/s in the bench
pipeline per slice count (vfm_debug_set_coarse_slices; 0 = the library's rule).  1920 workgroups (48 slices x 40 query blocks) are 7.5
rounds of 256 compute units: the last round is half empty."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import numpy as np
import torch
import bench
from vfmreg import _lib, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p) for p in range(4)]
q, b = pairs[0]["q_desc"], pairs[0]["b_desc"]
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, 24, st))
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
a, e = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(e))
for rep in range(2):
    for sl in [int(x) for x in (sys.argv[1:] or ["0", "32", "40", "44", "48", "51", "57", "64"])]:
        lib.vfm_debug_set_coarse_slices(sl)
        ts = []
        for r in range(14):
            lib.vfm_prof_arm(a, e)
            _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), 8, gate, st))
            torch.cuda.synchronize()
            ms = C.c_float(); lib.vfm_prof_elapsed_ms(a, e, C.byref(ms))
            if r >= 2: ts.append(ms.value)
        ts.sort()
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="mx6-half")
        v20, _, c20, _ = bench.timed_loop(lib, pipe, pairs, 20, 3)
        v200, _, c200, _ = bench.timed_loop(lib, pipe, pairs, 200, 3)
        del pipe
        print(f"slices {sl:2d}: coarse kernel alone {ts[len(ts)//2]:.3f} ms | pipeline 20 steps {v20:7.1f}/s (kernel {c20:.3f} ms), 200 steps {v200:7.1f}/s (kernel {c200:.3f} ms)", flush=True)
