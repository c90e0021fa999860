"""A/B timing of the coarse-kernel variants on config C2 (interleaved rounds, HIP events)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import _lib, ops, synth  # noqa: E402

lib = _lib.load()
import os  # noqa: E402
n, m, d = 20000, 200000, int(os.environ.get("D", "384"))
p = synth.make_pair_device(n, m, d, seed=42)
variants = [int(v) for v in sys.argv[1:]] or [1, 2]
a, b = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(b))
res = {v: [] for v in variants}
ref = None
for rnd in range(6):
    for v in variants:
        lib.vfm_debug_set_coarse_variant(v)
        lib.vfm_prof_arm(a, b)
        idx, sim = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST)
        ms = C.c_float()
        lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
        if rnd:
            res[v].append(ms.value)
        if ref is None:
            ref = idx.clone()
        assert torch.equal(idx, ref), f"variant {v} changed the result"
inl = p["match"] >= 0
print("planted recovered:", float((ref[inl] == p["match"][inl]).float().mean()))
for v in variants:
    t = sorted(res[v])
    print(f"variant {v}: median {t[len(t)//2]:.3f} ms  min {t[0]:.3f}  -> {2*n*m*d/(t[len(t)//2]*1e-3)/1e12:.0f} TFLOP/s")
