"""Row A6: vfm_match_mutual_pairs (find_correspondences' mutual filter in one call) and vfm_match_mutual_l2 (both full
directions) at C2 size on D.2 descriptors; milliseconds per call (HIP events).   python tools/time_pairs.py [reps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import ops, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for (n, m, d) in ((20000, 200000, 384), (5000, 50000, 384), (20000, 200000, 768)):
    p = synth.make_pair_device(n, m, d, seed=42)
    a, b = p["q_desc"], p["b_desc"]
    for name, fn in (("mutual_pairs", lambda: ops.match_mutual_pairs(a, b)), ("mutual_l2 (nn_ab + full nn_ba)", lambda: ops.match_mutual_l2(a, b)),
                     ("nn_ab only", lambda: ops.match_mutual_l2(a, b, mutual=False))):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{n} x {m} x {d} {name}: {sorted(ts[1:])[len(ts[1:]) // 2]:.2f} ms (min {min(ts[1:]):.2f})", flush=True)
