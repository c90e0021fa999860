"""Profiling driver: runs only the matching stage (prepare + coarse + select + rescore) of config C2
a few times, so that rocprofv3 --pmc passes focus on match_coarse_kernel.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/prof_match.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

import os  # noqa: E402

from vfmreg import _lib  # noqa: E402

if os.environ.get("VFM_LIB"):  # experimental build (tools/build_ablate.sh)
    _lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / os.environ["VFM_LIB"]
from vfmreg import ops, synth  # noqa: E402

_lib.load().vfm_debug_set_coarse_variant(int(os.environ.get("VFM_VARIANT", "0")))
_lib.load().vfm_debug_set_coarse_slices(int(os.environ.get("VFM_SLICES", "0")))

n, m, d = 20000, 200000, 384
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = synth.make_pair_device(n, m, d, seed=42)
lib = _lib.load()
records = int(os.environ.get("VFM_RECORDS", "0"))   # 0 = best-score records (what bench.py's pipeline runs), 1 = packed top-2
g = os.environ.get("VFM_GATE", "0.8")
if g == "none":   # the ungated one-shot call
    for _ in range(reps):
        idx, sim = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST)
else:             # the gated family, split form, as vfmreg/pipeline.py calls it
    q, b = p["q_desc"], p["b_desc"]
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
    idx = torch.empty(n, dtype=torch.int64, device="cuda")
    sim = torch.empty(n, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(reps):
        _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d,
                                                  (8 | 16) if records in (7, 8) else 8 if records in (5, 6, 9, 10) else 0, st))   # VFM_PREPARE_MX6 (| _MX6_HALF) for the VFM_RECORDS_MX6* kinds
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records, float(g), st))
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), float(g), records, st))
torch.cuda.synchronize()
print("ok", int((idx == p["match"]).sum()), "unresolved", int((idx < 0).sum()))  # meaningless for ablated builds
