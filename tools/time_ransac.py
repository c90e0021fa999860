"""vfm_ransac_corr alone at C2 / C3 correspondence counts (50 000 hypotheses): ms per call (HIP events), the fused 5-launch chain of
round 6 (default) against round 5's 11 launches (vfm_config key "ransac_fused")."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, ops, synth  # noqa: E402
for C in (10000, 19839):
    rng = np.random.default_rng(C)
    T = synth.random_pose(rng)
    src = np.c_[rng.uniform(-60, 60, C), rng.uniform(-60, 60, C), rng.uniform(-3, 12, C)]
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.02, src.shape)
    corres = np.stack([np.arange(C), np.arange(C)], 1).astype(np.int32)
    s, t, c = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (src, tgt, corres))
    for fused in (2, 1, 0, 2, 1, 0):
        _lib.thread_config().set("ransac_fused", fused)
        out = None
        ts = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = ops.ransac_corr(s, t, c, 10000.0, 50000, seed=42, out=out)
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b))
        name = {2: "chain 2 (8 launches)", 1: "chain 1 (5 launches)", 0: "round 5 chain (11 launches)"}[fused]
        print(f"C = {C}, {name}: {sorted(ts[2:])[len(ts[2:]) // 2]:.3f} ms per call, best hypothesis {int(out['best_hyp'].item())}, "
              f"rmse {float(out['rmse'].item()):.6f}")
_lib.thread_config().set("ransac_fused", 2)
