#!/usr/bin/env python
"""Where and when the workgroups of the token-stationary GEMM kernel run (its per-workgroup trace: start / end in 100 MHz ticks, XCC and
HW_ID): the last such launch of a one-block ViT forward (fc1), per number of images."""
import ctypes as C
import sys
from collections import Counter
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
all_imgs = torch.from_numpy(rng.integers(1, 255, (112, 1200, 1600, 3), dtype=np.uint8)).cuda()
buf = torch.zeros((1024, 4), dtype=torch.int64, device="cuda")
ptr = buf.data_ptr()
lib.vfm_debug_set_vit_gemm(-11, C.c_int32(ptr & 0xffffffff).value)
lib.vfm_debug_set_vit_gemm(-12, C.c_int32((ptr >> 32) & 0xffffffff).value)
lib.vfm_debug_set_vit_gemm(-9, 1)
for nimg in [int(x) for x in (sys.argv[1:] or ["88", "93", "94", "96", "112"])]:
    model = V.ViTS14(V.random_weights(0, depth=int(__import__("os").environ.get("VFM_DEPTH", "1"))), 1200, 1600)
    imgs = all_imgs[:nimg]
    for _ in range(3):
        buf.zero_()
        model.forward(imgs)
    torch.cuda.synchronize()
    groups = (nimg * ((16 * model.patch_w + 1 + 31) // 32 * 32) // 32 + 3) // 4
    t = buf[:groups].cpu().numpy()
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    xcc = (t[:, 2] >> 32) & 0xf
    hw = t[:, 2] & 0xffffffff
    cu = (xcc << 8) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xf)
    late = start > 10.0
    per_xcc = Counter(int(x) for x in xcc)
    dup = [c for c, k in Counter(int(x) for x in cu).items() if k > 1]
    print(f"{nimg} images, {groups} workgroups: kernel {end.max():.1f} us; durations min / median / max {np.min(end - start):.1f} / {np.median(end - start):.1f} / "
          f"{np.max(end - start):.1f} us; {int(late.sum())} started > 10 us after the first (at {np.round(np.sort(start[late])[:8], 1).tolist()}); "
          f"workgroups per XCC {[per_xcc.get(i, 0) for i in range(8)]}; compute units used twice: {len(dup)}; blockIdx % 8 == XCC for "
          f"{int((np.arange(groups) % 8 == xcc).sum())} of {groups}", flush=True)
    del model
