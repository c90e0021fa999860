#!/usr/bin/env python
"""The reference-shaped call with RANSAC's two-level path (bounds -> candidates -> exact) against scoring every hypothesis in fp64
(vfm_debug_set_ransac_exact_only): at the call's ~10^3 correspondences the second is fewer, shorter launches."""
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
for n_scan, n_map in ((20000, 200000), (60000, 200000)):
    p = synth.make_pair(n_scan, n_map, 384, seed=11)
    voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
    raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
    res = []
    for mode in (0, 1, 0, 1):
        lib.vfm_debug_set_ransac_exact_only(mode)
        node = RegistrationNode(cache_map=True)
        out = node.ransac_registration(voxel_map, raw_scan, "vfm")
        ts = []
        for _ in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = node.ransac_registration(voxel_map, raw_scan, "vfm")
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res.append((mode, sorted(ts)[len(ts) // 2] * 1e3, np.asarray(out[0])))
    print(f"ransac_registration scan {n_scan} / map {n_map}: " + ", ".join(f"exact-only {m}: {t:.3f} ms" for m, t, _ in res), flush=True)
    assert all(np.array_equal(res[0][2], r[2]) for r in res), "poses differ"
# RANSAC alone at a few correspondence counts
rng = np.random.default_rng(1)
for C in (300, 850, 2000, 4000, 9891):
    T = synth.random_pose(rng)
    src = rng.uniform(-40, 40, (C, 3))
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.02, src.shape)
    corr = np.stack([np.arange(C), np.arange(C)], 1).astype(np.int32)
    corr[::3, 1] = rng.integers(0, C, len(corr[::3]))
    s, t, c = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), torch.from_numpy(corr).cuda()
    row = []
    for mode in (0, 1):
        lib.vfm_debug_set_ransac_exact_only(mode)
        o = ops.ransac_corr(s, t, c, 0.5, 50000, seed=42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            o = ops.ransac_corr(s, t, c, 0.5, 50000, seed=42)
        torch.cuda.synchronize()
        row.append(((time.perf_counter() - t0) / 20 * 1e6, o["T"].cpu().numpy()))
    print(f"RANSAC alone, {C} correspondences, 50000 hypotheses: two-level {row[0][0]:.0f} us, every hypothesis in fp64 {row[1][0]:.0f} us, same pose: {np.array_equal(row[0][1], row[1][1])}", flush=True)
lib.vfm_debug_set_ransac_exact_only(0)
