#!/usr/bin/env python
"""Side stages of the registration pipeline on CU-masked streams (hipExtStreamCreateWithCUMask): the coarse kernel's
workgroups need a whole compute unit's registers, so a side kernel overlaps it only by taking compute units away from it;
memory-bound side kernels reach their bandwidth on a fraction of the chip.  C2 registrations per second by mask widths."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(ncu, offset=0, total=256):
    """A stream whose kernels may use `ncu` of the `total` compute units (mask bits offset .. offset + ncu - 1)."""
    words = (C.c_uint32 * (total // 32))()
    for i in range(offset, offset + ncu):
        words[(i % total) // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), total // 32, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


dev = torch.device("cuda:0")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + i, device=dev) for i in range(2)]


prep_cu, solve_cu = int(sys.argv[1]), int(sys.argv[2])   # 0 = every compute unit (torch's own stream)
pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2)
if prep_cu:
    pipe.prep_stream = masked_stream(prep_cu, 0)
if solve_cu:
    pipe.solve_streams = [masked_stream(solve_cu, prep_cu + i * solve_cu) for i in range(2)]
    pipe.ransac_stream = pipe.solve_streams[0]
times = []
for rnd in range(4):
    for i in range(12):
        p = pairs[i % 2]
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300):
        p = pairs[i % 2]
        pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
    pipe.synchronize()
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3 / 300)
times.sort()
print(json.dumps({"prepare_cus": prep_cu or "all", "solve_cus_each": solve_cu or "all", "ms_per_step_median": round(times[len(times) // 2], 4),
                  "min": round(times[0], 4), "per_s": round(1e3 / times[0], 1), "T00": float(out["T"][0, 0])}), flush=True)
