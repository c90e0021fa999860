#!/usr/bin/env python
"""The COLD reference-shaped call (RegistrationNode with cache_map=False, the default: registration_node.py:402-403 rebuilds the map from
the array in every call) step by step at C2 size (200 000 x 387 fp32 map): upload, voxel cap, container replay, gather, cast -- and the
whole call cold / through a set_map() handle."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import ops, synth  # noqa: E402
from vfmreg.mapping import VoxelHashMap  # noqa: E402
from vfmreg.registration import RegistrationNode  # noqa: E402
from vfmreg.voxelization import to_device_rows  # noqa: E402

VoxelHashMap.quiet = True
pp = synth.make_pair(20000, 200000, 384, seed=11)
voxel_map = np.c_[pp["b_xyz"], pp["b_desc"]].astype(np.float32)
raw_scan = np.c_[pp["q_xyz"], pp["q_desc"]].astype(np.float32)


def t(fn, reps=5):
    ts = []
    r = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return sorted(ts)[len(ts) // 2], r


ms, (rows, xyz) = t(lambda: to_device_rows(voxel_map))
print(f"to_device_rows (upload {voxel_map.nbytes / 1e6:.0f} MB + xyz as fp64): {ms:.2f} ms")
ms, keep = t(lambda: ops.voxel_first(xyz, 1.0, 20))
print(f"voxel_first (cap 20 per voxel): {ms:.2f} ms, kept {len(keep)}")
ms, kr = t(lambda: (rows[keep], xyz[keep]))
print(f"gather kept rows: {ms:.2f} ms")
ms, order = t(lambda: ops.voxel_robin(kr[1], 1.0, 20, reserve=False, hash_mul=ops.HASH_MAP))
print(f"voxel_robin (growing map, container order): {ms:.2f} ms")
ms, od = t(lambda: (kr[0][order], kr[1][order]))
print(f"gather in container order: {ms:.2f} ms")
ms, bd = t(lambda: od[0][:, 3:].float().contiguous())
print(f"descriptor columns as fp32, contiguous: {ms:.2f} ms")
ms, _ = t(lambda: ops.PreparedRows(bd))
print(f"search operand of the map (PreparedRows): {ms:.2f} ms")
node = RegistrationNode()
node.ransac_registration(voxel_map, raw_scan, "vfm")
ms, _ = t(lambda: node.ransac_registration(voxel_map, raw_scan, "vfm"))
print(f"whole call, cold (default node): {ms:.2f} ms")
ms, h = t(lambda: node.set_map(voxel_map), reps=3)
print(f"set_map: {ms:.2f} ms")
node.ransac_registration(h, raw_scan, "vfm")
ms, _ = t(lambda: node.ransac_registration(h, raw_scan, "vfm"), reps=15)
print(f"whole call through the handle: {ms:.3f} ms")
