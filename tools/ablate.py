"""Timing of ablated builds of the coarse kernel (results are garbage by construction)."""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import _lib, synth
name = sys.argv[1]
_lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / name
lib = _lib.load()
from vfmreg import ops
n, m, d = 20000, 200000, 384
p = synth.make_pair_device(n, m, d, seed=42)
a, b = C.c_void_p(), C.c_void_p()
lib.vfm_prof_events_create(C.byref(a), C.byref(b))
ts = []
for r in range(6):
    lib.vfm_prof_arm(a, b)
    ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST)
    ms = C.c_float(); lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
    if r: ts.append(ms.value)
ts.sort(); print(name, "median %.3f ms" % ts[len(ts)//2], "-> %.0f TFLOP/s" % (2*n*m*d/(ts[len(ts)//2]*1e-3)/1e12))
