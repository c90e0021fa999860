#!/usr/bin/env python
"""ViT-S/14 forward time against the number of images in one call (6 = the cameras of one RobotCar / NCLT scan):
what a caller that prepares several scans at once (PS:50-107 walks a whole sequence) gets per scan."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import vit as V  # noqa: E402

FLOP_PER_IMAGE = None
rng = np.random.default_rng(0)
res = {}
all_imgs = torch.from_numpy(rng.integers(1, 255, (96, 1200, 1600, 3), dtype=np.uint8)).cuda()
from vfmreg import _lib  # noqa: E402
lib = _lib.load()
import os
if os.environ.get("VFM_VIT_LDS_SHAPE"):
    lib.vfm_debug_set_vit_gemm(-6, int(os.environ["VFM_VIT_LDS_SHAPE"]))
THRS = [int(x) for x in os.environ.get("VFM_VIT_LDS_THR", "0,512,256,128").split(",")]   # workgroups from which the LDS-tiled GEMM is used (0: never)
for thr, nimg in [(t, n) for t in THRS for n in (6, 12, 24, 48, 90, 96)]:
    lib.vfm_debug_set_vit_gemm(-5, thr)
    imgs = all_imgs[:nimg]
    model = V.ViTS14(V.random_weights(0), 1200, 1600)
    out = model.forward(imgs)
    torch.cuda.synchronize()
    if nimg == 6 and thr == THRS[0]:
        first = out.clone()
    else:
        assert torch.equal(out[:6], first), "a larger batch must not change the first scan's features"
    ts = []
    for _ in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        model.forward(imgs)
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    ms = ts[len(ts) // 2]
    tok = 337
    flop = nimg * 12 * (tok * 2 * 384 * (1152 + 384 + 2 * 1536) + 6 * 2 * 2 * tok * tok * 64) + nimg * 336 * 2 * 588 * 384
    res[f"{thr}:{nimg}"] = {"lds_from_workgroups": thr, "images": nimg, "ms": ms, "ms_per_scan_of_6": ms * 6 / nimg, "TFLOP_per_s": flop / ms / 1e9}
    print(json.dumps(res[f"{thr}:{nimg}"]), flush=True)
    del imgs, model, out
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "time_vit_batch.json").write_text(json.dumps(res, indent=1))
