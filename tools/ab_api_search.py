#!/usr/bin/env python
"""VoxelHashMap.search_device on a kept map, ~10^3 queries (the reference-shaped call's search): half-width coarse pass
(VFM_RECORDS_HALF) against best-score records at full width, alternating on the same box."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import synth  # noqa: E402
from vfmreg.mapping import VoxelHashMap  # noqa: E402

VoxelHashMap.quiet = True
for nq, m in ((800, 200000), (1700, 200000), (1700, 100000), (4000, 200000)):
    p = synth.make_pair(nq, m, 384, seed=5)
    vm = VoxelHashMap(0.01, 1.0e9, 20)
    vm.add_points(np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32))
    qd = torch.from_numpy(p["q_desc"].astype(np.float32)).cuda()
    vm.search_device(None, 0.8, q_desc=qd)
    probed = vm._half
    res = {}
    for rep in range(3):
        for half in (False, True):
            vm._half = half
            ts = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                qi, mi, _ = vm.search_device(None, 0.8, q_desc=qd)
                ts.append(time.perf_counter() - t0)
                vm._half = half
            res.setdefault(half, []).append(sorted(ts)[len(ts) // 2] * 1e6)
    print(f"{nq} queries x {m} rows (probe: half-width {'yes' if probed else 'no'}): full width {min(res[False]):.0f} us, half width {min(res[True]):.0f} us, "
          f"{len(qi)} correspondences", flush=True)
