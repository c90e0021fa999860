"""Timing of the mutual-NN L2 matcher (row A6) at a few sizes."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import ops
for (n, m, d) in ((5000, 5000, 33), (5000, 50000, 384), (20000, 200000, 384)):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(n, d, device="cuda", generator=g)
    b = torch.randn(m, d, device="cuda", generator=g)
    if len(sys.argv) > 1 and (n * m * d > float(sys.argv[1])):
        continue
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nn_ab, d2, nn_ba = ops.match_mutual_l2(a, b)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(n, m, d, "mutual L2: %.2f ms  (%.1f TFLOP/s on 4*n*m*d)" % (t * 1e3, 4.0 * n * m * d / t / 1e12))
for (n, m, d) in ((20000, 200000, 768),):
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(n, d, device="cuda", generator=g); b = torch.randn(m, d, device="cuda", generator=g)
    for r in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ops.match_mutual_l2(a, b)
        torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(n, m, d, "mutual L2 (row-bias form): %.2f ms  (%.1f TFLOP/s on 4*n*m*d)" % (t * 1e3, 4.0 * n * m * d / t / 1e12))
