"""Timing of the F rows at C2-like sizes: voxel down-sampling / first-K map build (F1) and ICP refinement (F2)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch
from vfmreg import synth
from vfmreg.voxelization import voxel_down_sample
from vfmreg.mapping import VoxelHashMap
from vfmreg.icp import register_frame

rng = np.random.default_rng(0)
m, n = 200000, 20000
b_xyz = np.c_[rng.uniform(-60, 60, m), rng.uniform(-60, 60, m), rng.uniform(-3, 12, m)]
T = synth.random_pose(rng)
T_small = np.eye(4); a = np.deg2rad(1.0)
T_small[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]; T_small[:3, 3] = [0.2, -0.1, 0.05]
scan = (b_xyz[rng.permutation(m)[:n]] - T_small[:3, 3]) @ T_small[:3, :3] + rng.normal(0, 0.02, (n, 3))


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r

t, ds = timed(lambda: voxel_down_sample(b_xyz, 1.0))
print(f"voxel_down_sample 200k pts, 1.0 m: {t:.2f} ms (host numpy in/out) -> {len(ds)} pts")
t, ds2 = timed(lambda: voxel_down_sample(scan, 0.5))
print(f"voxel_down_sample 20k pts, 0.5 m: {t:.2f} ms -> {len(ds2)} pts")
def build():
    vm = VoxelHashMap(1.0, 100.0, 20)
    vm.add_points(b_xyz)
    return vm
t, vm = timed(build, reps=3)
print(f"VoxelHashMap.add_points 200k pts (20 / voxel): {t:.2f} ms -> {len(vm.point_cloud())} pts")
t, pose = timed(lambda: register_frame(scan, vm, np.eye(4), 3.0, 1.0))
print(f"register_frame 20k scan vs map: {t:.2f} ms; pose err {np.linalg.norm(pose - T_small):.2e}")
