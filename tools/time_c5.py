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
ts = []
for r in range(4):
    lib.vfm_prof_arm(a, b)
    ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST)
    ms = C.c_float(); lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
    if r: ts.append(ms.value)
ts.sort(); t = ts[len(ts) // 2]
print(f"C5 coarse pass {n} x {m} x {d}: {t:.2f} ms -> {2 * n * m * d / (t * 1e-3) / 1e12:.0f} TFLOP/s")
pipe = RegistrationPipeline(n, m, d, n_iter=50000)
ts = []
for r in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort()
print(f"C5 registration (prepare + match + threshold + 50k-iteration RANSAC): {ts[1] * 1e3:.1f} ms; "
      f"{int(out['count'].item())} correspondences")
