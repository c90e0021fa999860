#!/usr/bin/env python
"""Where the reference-shaped call spends its time: the steps of RegistrationNode._correspond + the RANSAC call of
ransac_registration (vfmreg/registration.py), run by hand with a host time stamp behind each step that ends in a read-back."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

from vfmreg import o3d, ops, synth  # noqa: E402
from vfmreg.mapping import VoxelHashMap  # noqa: E402
from vfmreg.registration import RegistrationNode  # noqa: E402

VoxelHashMap.quiet = True
n_scan, n_map = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 200000)
p = synth.make_pair(n_scan, n_map, 384, seed=11)
voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
node = RegistrationNode(cache_map=True)
node.ransac_registration(voxel_map, raw_scan, "vfm")
vs = node.config.mapping.voxel_size
acc = {}


def run():
    marks = []

    def mark(name):
        marks.append((name, time.perf_counter()))

    torch.cuda.synchronize()
    mark("start")
    scan = np.asarray(raw_scan)
    xyz_h = np.ascontiguousarray(scan[:, :3], dtype=np.float64)
    mark("host: xyz columns as fp64")
    xyz = torch.from_numpy(xyz_h).cuda()
    mark("upload xyz")
    pose = np.ascontiguousarray(np.eye(4), dtype=np.float64)
    key = pose.tobytes()
    if node._pose_cache is None or node._pose_cache[0] != key:
        node._pose_cache = (key, torch.from_numpy(pose).cuda())
    T = node._pose_cache[1]
    chain = node._voxel_chain(xyz, T, vs)
    mark("three chained voxelisations: three launches, one read-back (levels' sizes + the 5 m level's rows)")
    xyz, raw_of, order, raw_idx = chain
    vhm = node._hash_map_for(voxel_map)
    pcl = ops.transform_xyz(xyz, T)
    sub = pcl[order]
    mark("gather of voxel_scan, map lookup (fingerprint), transform, index")
    q_desc = node._upload_rows(scan, raw_idx)
    torch.cuda.current_stream().synchronize()   # (measurement only: the copy is asynchronous)
    mark("host: gather descriptor rows + upload")
    qi, mi, _ = vhm.search_device(None, node.min_cosine_similarity, q_desc=q_desc)
    mark("search_device (prepare, search, compact, count)")
    src_rows, tgt_rows = order[qi], mi
    pcd_src = o3d.geometry.PointCloud()
    pcd_src.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(xyz))
    pcd_tgt = o3d.geometry.PointCloud()
    pcd_tgt.points = o3d.utility.Vector3dVector(o3d.utility.DeviceArray(vhm.point_cloud_device()))
    coors = o3d.utility.Vector2iVector(o3d.utility.DeviceArray(torch.stack((src_rows, tgt_rows), dim=1).to(torch.int32)))
    mark("index, stack, wrappers")
    res = o3d.pipelines.registration.registration_ransac_based_on_correspondence(
        pcd_src, pcd_tgt, coors, node.max_correspondence_distance,
        o3d.pipelines.registration.TransformationEstimationPointToPoint(False), ransac_n=3,
        criteria=o3d.pipelines.registration.RANSACConvergenceCriteria(node.ransac_iterations, 1))
    pose = np.array(res.transformation)
    mark("ransac (bounds read-back, RANSAC, result read-back)")
    for (a, ta), (b, tb) in zip(marks[:-1], marks[1:]):
        acc.setdefault(b, []).append(tb - ta)
    acc.setdefault("total", []).append(marks[-1][1] - marks[0][1])
    return pose


for _ in range(3):
    run()
acc.clear()
for _ in range(20):
    run()
print(f"scan {n_scan} / map {n_map}, median of 20 calls, microseconds:")
for k, v in acc.items():
    print(f"  {np.median(v) * 1e6:8.1f}  {k}")
