#!/usr/bin/env python
"""Phase by phase through the one-launch VoxelDownsample kernel (voxel_robin_grid_kernel): wall-clock stamps of workgroup 0 behind
every grid-wide barrier (vfm_debug_voxel_trace), microseconds between them."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import _lib, ops  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(3)
names = ["0 tables", "A insert", "B count", "B write + hist", "C sum", "C offsets", "C scatter", "D max", "D flags", "D clusters (+rot)",
         "E replay", "F emit"]
lib.vfm_debug_set_voxel_small(101)
for n, ppt in ((1700, 1), (20000, 1), (60000, 1), (60000, 4), (200000, 1)):
    lib.vfm_debug_set_voxel_small(10 + ppt)
    d = torch.from_numpy(rng.uniform(-60, 60, (n, 3)) * [1, 1, 0.15]).cuda()
    keep = torch.empty(n, dtype=torch.int64, device="cuda")
    count = torch.empty(1, dtype=torch.int64, device="cuda")
    info = (C.c_int64 * 4)()
    ws = torch.empty(lib.vfm_voxel_robin_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    acc = []
    for rep in range(6):
        _lib.check(lib.vfm_voxel_robin(d.data_ptr(), n, 3, 0.5, 1, ops.HASH_DOWNSAMPLE, n, keep.data_ptr(), count.data_ptr(),
                                       C.cast(info, C.c_void_p), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "voxel_robin")
        out = (C.c_int64 * 32)()
        _lib.check(lib.vfm_debug_voxel_trace(ws.data_ptr(), n, C.cast(out, C.c_void_p)), "trace")
        k = int(out[31])
        acc.append([(out[i + 1] - out[i]) / 100.0 for i in range(k - 1)])
    a = np.median(np.array(acc[1:]), axis=0)
    print(f"n = {n}, {ppt} point(s) per thread: total {a.sum():.1f} us: " + ", ".join(f"{nm} {v:.1f}" for nm, v in zip(names, a)), flush=True)
lib.vfm_debug_set_voxel_small(100)
lib.vfm_debug_set_voxel_small(10)
