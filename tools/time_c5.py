"""Config C5 (stretch): coarse pass and whole registration at 50 000 x 1 000 000 x 768."""
import ctypes as C, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import _lib, ops, synth
from vfmreg.pipeline import RegistrationPipeline
lib = _lib.load()
n, m, d = 50000, 1000000, 768
p = synth.make_pair_device(n, m, d, seed=1)
a, b = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(b))
for gate, label in ((None, "fp16 pass (ungated call)"), (0.8, "int8 pass (gated call)")):
    ts, tt = [], []
    for r in range(4):
        lib.vfm_prof_arm(a, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx, sim = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST, gate=gate)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ms = C.c_float(); lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
        if r: ts.append(ms.value); tt.append(1e3 * (t1 - t0))
    ts.sort(); t = ts[len(ts) // 2]
    ok = idx >= 0
    print(f"C5 {label}: coarse kernel {n} x {m} x {d}: {t:.2f} ms -> {2 * n * m * d / (t * 1e-3) / 1e12:.0f} T/s; whole search "
          f"{sorted(tt)[len(tt) // 2]:.1f} ms; planted matches recovered {int((idx[ok] == p['match'][ok]).sum())} of {int((p['match'] >= 0).sum())}", flush=True)
pipe = RegistrationPipeline(n, m, d, n_iter=50000)
ts = []
for r in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"C5 registration (prepare + match + threshold + 50k-iteration RANSAC): {ts[1] * 1e3:.1f} ms; "
      f"{int(out['count'].item())} correspondences")
