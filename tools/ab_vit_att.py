"""ViT-S/14 forward with the attention's K / V^T fetched per wave from the L2 (vit_attention_kernel) against staged once per workgroup in
the LDS (vit_attention_lds_kernel): 6 / 24 / 96 images per call."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib  # noqa: E402
from vfmreg import vit as V  # noqa: E402
lib = _lib.load()
rng = np.random.default_rng(0)
all_imgs = torch.from_numpy(rng.integers(1, 255, (96, 1200, 1600, 3), dtype=np.uint8)).cuda()
for rep in range(2):
    for nimg in (6, 24, 96):
        for mode in (0, 1):
            lib.vfm_debug_set_vit_gemm(-7, mode)
            model = V.ViTS14(V.random_weights(0), 1200, 1600)
            imgs = all_imgs[:nimg]
            model.forward(imgs)
            torch.cuda.synchronize()
            ts = []
            for _ in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                model.forward(imgs)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            print(f"{nimg:3d} images, attention K / V^T {'in the LDS' if mode else 'per wave from the L2'}: {sorted(ts)[6]:.3f} ms", flush=True)
            del model
lib.vfm_debug_set_vit_gemm(-7, 1)
