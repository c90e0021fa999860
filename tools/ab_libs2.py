"""A/B of two builds of the library: the coarse kernels alone (half-width, best-score, top-2 records, fp6) and the bench's
pipeline (auto / int8 / mx6, 20 and 200 steps); each build in its own process, alternating, on the same box.
    python tools/ab_libs2.py libvfmreg_hip_base.so libvfmreg_hip.so"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys
from pathlib import Path
ROOT = Path(sys.argv[1]); sys.path.insert(0, str(ROOT / "vfm-registration_amd")); sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np, torch
from vfmreg import _lib
_lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / sys.argv[2]
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
import bench, dev_mx6 as D
lib = _lib.load()
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + j) for j in range(2)]
q, b = pairs[0]["q_desc"], pairs[0]["b_desc"]
gate = float(np.nextafter(np.float32(0.8), np.float32(-np.inf)))
has6 = hasattr(lib, "vfm_debug_mx6_rows")
qb, bb = D.prepare(b, q, 8 if has6 else 0)
out = []
for rec in (0, 1) + ((5,) if has6 else ()):   # (operands prepared with the fp6 image carry no half-width one)
    ts = sorted(D.search(q, b, qb, bb, gate, rec)[2] for _ in range(9))
    out.append(f"rec{rec} {ts[4]:.3f}")
for mode in ("auto", "int8") + (("mx6",) if has6 else ()):
    for steps in (20, 200):
        pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
        v, msps, cms, res = bench.timed_loop(lib, pipe, pairs, steps, 3, settle=4 if mode == "auto" else 0)
        out.append(f"{mode}{steps} {v:.0f}/s({cms:.3f})")
        del pipe
print(" | ".join(out))
'''
libs = sys.argv[1:]
for rep in range(3):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), l], capture_output=True, text=True)
        print(l, "::", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
