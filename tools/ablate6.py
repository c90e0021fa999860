"""Timing of ablated builds of the fp6 coarse kernel (tools/build_ablate6.sh; results are garbage by construction):
python tools/ablate6.py libvfmreg_hip_NAME.so [records ...]   -> median coarse-kernel time at C2 per record kind"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import numpy as np
import torch
from vfmreg import _lib, synth
name = sys.argv[1]
kinds = [int(x) for x in sys.argv[2:]] or [8, 5]
_lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / name
lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
q, b = p["q_desc"], p["b_desc"]
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=torch.uint8, device="cuda")
_lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, 8, st))
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
a, e = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(e))
out = []
for rec in kinds:
    ts = []
    for r in range(12):
        lib.vfm_prof_arm(a, e)
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), rec, gate, st))
        torch.cuda.synchronize()
        ms = C.c_float(); lib.vfm_prof_elapsed_ms(a, e, C.byref(ms))
        if r >= 2: ts.append(ms.value)
    ts.sort()
    kc = d // 2 if rec in (7, 8) else d
    out.append(f"records {rec}: {ts[len(ts)//2]:.3f} ms ({2*n*m*kc/(ts[len(ts)//2]*1e-3)/1e15:.2f} PFLOP/s)")
print(f"{name:36s}", " | ".join(out), flush=True)
