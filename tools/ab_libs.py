"""A/B of two builds of the library on the pipelined registration throughput: each build runs in its own
process (the .so cannot be swapped in-process), alternating, on the same box.
    python tools/ab_libs.py libvfmreg_hip.so libvfmreg_hip_X.so"""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, time, os
from pathlib import Path
ROOT = Path(sys.argv[1]); sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import _lib
_lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / sys.argv[2]
from vfmreg import synth
from vfmreg.pipeline import RegistrationPipeline
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + j) for j in range(2)]
torch.cuda.synchronize(); ready = torch.cuda.Event(); ready.record()
pipe = RegistrationPipeline(n, m, d, overlap_ransac=True)
res = []
for rnd in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30):
        p = pairs[i % 2]
        pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ready)
    pipe.synchronize(); torch.cuda.synchronize()
    if rnd: res.append(30 / (time.perf_counter() - t0))
res.sort(); print(res[len(res)//2])
'''
libs = sys.argv[1:]
out = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), l], capture_output=True, text=True)
        out[l].append(float(r.stdout.strip().splitlines()[-1]))
for l in libs:
    print(l, " ".join(f"{v:.1f}" for v in out[l]))
