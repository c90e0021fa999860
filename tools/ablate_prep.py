"""Timing of ablated builds of the operand-preparation kernel (tools/build_ablate_prep.sh; results are garbage by construction):
python tools/ablate_prep.py libvfmreg_hip_NAME.so [flags ...]   -> time of vfm_match_prepare2_gated_p at C2 per flag set"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import _lib, synth
name = sys.argv[1]
flag_sets = [int(x) for x in sys.argv[2:]] or [24, 8, 0]
_lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / name
lib = _lib.load()
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
q, b = p["q_desc"], p["b_desc"]
st = torch.cuda.current_stream().cuda_stream
qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=torch.uint8, device="cuda")
bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=torch.uint8, device="cuda")
out = []
for flags in flag_sets:
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    out.append(f"flags {flags}: {ts[len(ts) // 2]:.3f} ms")
print(f"{name:36s}", " | ".join(out), flush=True)
