#!/usr/bin/env python
"""The reference-shaped call the node makes -- RegistrationNode.ransac_registration(voxel_map, raw_scan, 'vfm', run_icp=True)
(registration_node.py:273-357: three chained voxelisations, hash-map build, descriptor search, index recovery, 50 000-iteration
RANSAC, ICP refinement; numpy in, numpy out) -- timed on the GPU build and on the CPU oracle's restatement of the same steps."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from oracle import oracle as orc  # noqa: E402  (timing comparison only)
from vfmreg import synth  # noqa: E402
from vfmreg.mapping import VoxelHashMap  # noqa: E402
from vfmreg.registration import RegistrationNode  # noqa: E402

VoxelHashMap.quiet = True
def timed(node, voxel_map, raw_scan, icp, reps=5):
    node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=icp)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=icp)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], out


for n_scan, n_map in ((6000, 30000), (20000, 100000), (20000, 200000), (60000, 200000)):
    p = synth.make_pair(n_scan, n_map, 384, seed=11)
    voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
    raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
    # round 4: the map of a scene is built once and kept across its scans (RegistrationNode(cache_map=True));
    # cache_map=False is round 3's behaviour (upload + container replay of the map in every call)
    t_cold, _ = timed(RegistrationNode(cache_map=False), voxel_map, raw_scan, True, reps=3)
    t_noicp, _ = timed(RegistrationNode(cache_map=True), voxel_map, raw_scan, False)
    t_gpu, (pose, pose_icp) = timed(RegistrationNode(cache_map=True), voxel_map, raw_scan, True)
    print(f"scan {n_scan} / map {n_map}: map rebuilt every call {1e3 * t_cold:.1f} ms; map kept: {1e3 * t_noicp:.1f} ms without ICP, "
          f"{1e3 * t_gpu:.1f} ms with", flush=True)
    t0 = time.perf_counter()
    ref_pose, ref_icp, corres = orc.ransac_registration_vfm(voxel_map, raw_scan, n_iter=50000, run_icp=True)
    t_cpu = time.perf_counter() - t0
    same = bool(np.array_equal(pose, ref_pose) and np.array_equal(pose_icp, ref_icp))
    print(f"scan {n_scan} / map {n_map} rows x 387: GPU build {1e3 * t_gpu:.1f} ms, CPU oracle ({orc.num_threads()} threads) "
          f"{1e3 * t_cpu:.0f} ms, {len(corres)} correspondences, poses bit-equal: {same}, "
          f"pose err vs planted {np.linalg.norm(pose_icp - p['T_gt']):.4f}", flush=True)
