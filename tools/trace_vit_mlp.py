#!/usr/bin/env python
"""Where vit_mlp_kernel's workgroups spend their time (its trace: start, loop start, loop end, end in 100 MHz ticks; shader clocks at the
loop's waits + barriers), last layer of a forward.  python tools/trace_vit_mlp.py [nimg ...]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, vit as V  # noqa: E402
lib = _lib.load()
rng = np.random.default_rng(0)
buf = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
ptr = buf.data_ptr()
lib.vfm_debug_set_vit_gemm(-11, C.c_int32(ptr & 0xffffffff).value)
lib.vfm_debug_set_vit_gemm(-12, C.c_int32((ptr >> 32) & 0xffffffff).value)
_lib.thread_config().set("vit_fused_mlp", 1)
_lib.thread_config().set("vit_trace_fused", 3)
model = V.ViTS14(V.random_weights(0, depth=2), 1200, 1600)
for nimg in [int(x) for x in (sys.argv[1:] or ["6", "84"])]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    for _ in range(3):
        buf.zero_()
        model.forward(imgs)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t[:, :4] - t0) / 100.0
    print(f"{nimg} images, {len(t)} workgroups, kernel {rel[:, 3].max():.1f} us; start max {rel[:, 0].max():.1f}; prologue {np.median(rel[:, 1] - rel[:, 0]):.2f} us, "
          f"loop {np.median(rel[:, 2] - rel[:, 1]):.2f} us ({np.median(rel[:, 2] - rel[:, 1]) / 48:.3f} per iteration), tail + epilogue {np.median(rel[:, 3] - rel[:, 2]):.2f} us; "
          f"shader clocks at waits + barriers {np.median(t[:, 4]) / 48:.0f} per iteration; loop {np.median(t[:, 5]) / 48:.0f} counter ticks per iteration "
          f"= {np.median(t[:, 5]) / np.median(rel[:, 2] - rel[:, 1]) / 1e3:.3f} GHz if the counter is the shader clock")
