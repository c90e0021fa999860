#!/usr/bin/env python
"""VoxelDownsample through vfm_voxel_robin: the general multi-launch path (vfm_debug_set_voxel_small(0)) against the one-launch
cooperative kernel (1, default), per call and inside the reference-shaped registration call (tools/time_api.py's workload)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import _lib, ops, synth  # noqa: E402
from vfmreg.mapping import VoxelHashMap  # noqa: E402
from vfmreg.registration import RegistrationNode  # noqa: E402

VoxelHashMap.quiet = True
lib = _lib.load()
rng = np.random.default_rng(3)
for n in (1700, 6000, 20000, 60000, 200000):
    d = torch.from_numpy(rng.uniform(-60, 60, (n, 3)) * [1, 1, 0.15]).cuda()
    row = []
    for mode in (0, 1, 12, 14, 18):
        lib.vfm_debug_set_voxel_small(1 if mode else 0)
        lib.vfm_debug_set_voxel_small(mode if mode > 10 else 10)
        outs = ops.voxel_robin(d, 0.5)
        ts = []
        for _ in range(30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            o = ops.voxel_robin(d, 0.5)
            ts.append(time.perf_counter() - t0)
        row.append((sorted(ts)[len(ts) // 2] * 1e3, int(o.numel())))
        assert torch.equal(o, outs)
    print(f"voxel_robin n = {n}: general path {row[0][0]:.3f} ms, one launch {row[1][0]:.3f} ms (points per thread by size); 2 / 4 / 8 points per thread "
          f"{row[2][0]:.3f} / {row[3][0]:.3f} / {row[4][0]:.3f} ({row[1][1]} voxels)", flush=True)
lib.vfm_debug_set_voxel_small(10)

for n_scan, n_map in ((20000, 200000), (60000, 200000)):
    p = synth.make_pair(n_scan, n_map, 384, seed=11)
    voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
    raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
    res = []
    for mode in (0, 1, 0, 1):
        lib.vfm_debug_set_voxel_small(mode)
        node = RegistrationNode(cache_map=True)
        for icp in (False,):
            out = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=icp)
            ts = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=icp)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            res.append((mode, sorted(ts)[len(ts) // 2] * 1e3, np.asarray(out[0] if isinstance(out, tuple) else out)))
    print(f"ransac_registration scan {n_scan} / map {n_map}: " + ", ".join(f"mode {m}: {t:.3f} ms" for m, t, _ in res), flush=True)
    assert all(np.array_equal(res[0][2], r[2]) for r in res), "poses differ"
lib.vfm_debug_set_voxel_small(1)
