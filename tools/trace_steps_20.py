#!/usr/bin/env python
"""The driver's form (3 warm-up + 20 timed registrations between two synchronisations) step by step: the interval between the starts of
consecutive coarse kernels (HIP events on the coarse stream), the first start after the timed region opens, the end of the region after the
last coarse kernel -- where the 20-step figure loses against the 200-step one."""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
n, m, d = 20000, 200000, 384
pairs = [synth.make_pair_device(n, m, d, seed=42 + p, device=dev) for p in range(2)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(3):
    pipe = RegistrationPipeline(n, m, d, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse="auto")
    main = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(main)
    reg = lambda i: pipe.register(pairs[i % 2]["q_desc"], pairs[i % 2]["q_xyz"], pairs[i % 2]["b_desc"], pairs[i % 2]["b_xyz"], want_mask=True, inputs_ready=ready)
    for i in range(4):
        reg(i)
        pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
    ev = []
    for _ in range(steps):
        a, b = C.c_void_p(), C.c_void_p()
        lib.vfm_prof_events_create(C.byref(a), C.byref(b))
        ev.append((a, b))
    open_ev = torch.cuda.Event(enable_timing=True)
    for i in range(3):
        reg(i)
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for i in range(steps):
        lib.vfm_prof_arm(ev[i][0], ev[i][1])
        reg(i)
        host.append((time.perf_counter() - t0) * 1e3)
    t_enq = time.perf_counter() - t0
    pipe.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    ms = C.c_float()
    gaps, durs = [], []
    for i in range(steps):
        lib.vfm_prof_elapsed_ms(ev[i][0], ev[i][1], C.byref(ms))
        durs.append(ms.value)
        if i + 1 < steps:
            lib.vfm_prof_elapsed_ms(ev[i][0], ev[i + 1][0], C.byref(ms))
            gaps.append(ms.value)
    span = sum(gaps) + durs[-1]
    print(f"run {rep}: {steps} steps in {dt:.3f} ms = {steps / dt * 1e3:.0f}/s; enqueued after {t_enq * 1e3:.3f} ms; first coarse start .. last coarse end {span:.3f} ms "
          f"-> {dt - span:.3f} ms outside it (fill in front + drain behind)")
    print("   start-to-start of consecutive coarse kernels: " + " ".join(f"{g:.3f}" for g in gaps))
    print("   coarse kernel durations:                      " + " ".join(f"{g:.3f}" for g in durs))
    print("   host: register() i returned at (ms):          " + " ".join(f"{h:.2f}" for h in host))
    del pipe
