"""ViT forward only, for rocprofv3: python tools/prof_vit.py [default_tiles=1|0 (0 = round-1 tiles)] [reps]"""
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
lib = _lib.load()
if not fused:
    lib.vfm_debug_set_vit_gemm(108, 208)
rng = np.random.default_rng(0)
imgs = torch.from_numpy(rng.integers(1, 255, (6, 1200, 1600, 3), dtype=np.uint8)).cuda()
model = V.ViTS14(V.random_weights(0), 1200, 1600)
for _ in range(reps):
    out = model.forward(imgs)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
