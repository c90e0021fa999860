"""ViT-S/14, one scan (6 images) and two: waves per workgroup of the direct GEMM kernel (vfm_debug_set_vit_gemm(-8, n); 0 = the library's rule)."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, vit as V  # noqa: E402
lib = _lib.load()
rng = np.random.default_rng(0)
imgs_all = torch.from_numpy(rng.integers(1, 255, (24, 1200, 1600, 3), dtype=np.uint8)).cuda()
first = {}
for rep in range(2):
    for wpw in (4, 2, 1, 0):
        lib.vfm_debug_set_vit_gemm(-8, wpw)
        for nimg in (6, 12, 24):
            imgs = imgs_all[:nimg]
            model = V.ViTS14(V.random_weights(0), 1200, 1600)
            out = model.forward(imgs)
            torch.cuda.synchronize()
            if nimg not in first:
                first[nimg] = out.clone()
            assert torch.equal(out, first[nimg])
            ts = []
            for _ in range(30):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                model.forward(imgs)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b))
            ts.sort()
            print(f"waves per workgroup {wpw}: {nimg:2d} images {ts[len(ts) // 2]:.4f} ms (min {ts[0]:.4f})", flush=True)
lib.vfm_debug_set_vit_gemm(-8, 0)
