"""ViT forward only, for rocprofv3: python tools/prof_vit.py [default_tiles=1|0 (0 = round-1 tiles)] [reps] [images] [lds threshold]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402

fused = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nimg = int(sys.argv[3]) if len(sys.argv) > 3 else 6        # images per call
lds_thr = int(sys.argv[4]) if len(sys.argv) > 4 else -1    # workgroups from which the LDS-tiled GEMM is used (-1: the library's default)
lib = _lib.load()
if not fused:
    lib.vfm_debug_set_vit_gemm(108, 208)
if lds_thr >= 0:
    lib.vfm_debug_set_vit_gemm(-5, lds_thr)
if len(sys.argv) > 5:                                      # token-stationary QKV / fc1 kernel from this many groups of 128 rows on (0: the default policy, -1: never)
    lib.vfm_debug_set_vit_gemm(-9, int(sys.argv[5]))
import os
if os.environ.get("VFM_HOT_A") == "1":                      # timing experiment (wrong results): tools/ab_vit_hot_a.sh
    lib.vfm_debug_set_vit_gemm(-16, 1)
if os.environ.get("VFM_FUSED_QKV"):                         # QKV + attention in one workgroup from this many images on (round 6)
    _lib.thread_config().set("vit_fused_qkv", int(os.environ["VFM_FUSED_QKV"]))
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
model = V.ViTS14(V.random_weights(0), 1200, 1600)
for _ in range(reps):
    out = model.forward(imgs)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
