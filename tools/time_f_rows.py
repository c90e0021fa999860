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

# ---- the device entry points alone (device-resident inputs; HIP events), with their compulsory bytes against 8 TB/s
from vfmreg import ops, _lib  # noqa: E402
from vfmreg.icp import VoxelGridDevice  # noqa: E402


def dev_ms(fn, reps=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def line(name, ms, nbytes):
    gbs = nbytes / (ms * 1e-3) / 1e9
    print(f"{name}: {ms:.3f} ms, compulsory {nbytes / 1e6:.1f} MB -> {gbs:.0f} GB/s = {gbs / 8000:.3f} of the 8 TB/s HBM peak")


bx = torch.from_numpy(b_xyz).cuda()
sx = torch.from_numpy(np.ascontiguousarray(scan)).cuda()
k1 = ops.voxel_first(bx, 1.0, 1)
line("vfm_voxel_first 200k pts, 1.0 m (K = 1; incl. the count read-back of the wrapper)", dev_ms(lambda: ops.voxel_first(bx, 1.0, 1)), m * 24 + len(k1) * 8)
line("vfm_voxel_first 200k pts, 1.0 m (K = 20)", dev_ms(lambda: ops.voxel_first(bx, 1.0, 20)), m * 24 + m * 8)
line("vfm_voxel_robin 200k pts, 1.0 m (reserved container: VoxelDownsample's order; synchronises by design)",
     dev_ms(lambda: ops.voxel_robin(bx, 1.0, 1, reserve=True)), m * 24 + len(k1) * 8)
line("vfm_voxel_robin 200k pts (growing map, K = 20: VoxelHashMap order)",
     dev_ms(lambda: ops.voxel_robin(bx, 1.0, 20, reserve=False, hash_mul=ops.HASH_MAP)), m * 24 + m * 8)
g = VoxelGridDevice(vm.point_cloud(), 1.0)
lib = _lib.load()
tgt = torch.empty_like(sx)
valid = torch.empty(n, dtype=torch.uint8, device="cuda")
out = torch.empty(43, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
eye = np.eye(4)
src2 = torch.empty_like(sx)
line("vfm_icp_step_nearest 20k pts (transform + 27-voxel nearest neighbour)",
     dev_ms(lambda: lib.vfm_icp_step_nearest(sx.data_ptr(), n, eye.ctypes.data, src2.data_ptr(), g.keys.data_ptr(), g.start.data_ptr(),
                                             g.pts.data_ptr(), g.n_voxels, 1.0, 3.0, tgt.data_ptr(), valid.data_ptr(), st)),
     n * (24 + 24 + 24 + 1) + n * 27 * (8 + 1.6 * 24))
line("vfm_icp_build_system 20k pairs (6x6 normal equations, fixed reduction tree, one workgroup per entry)",
     dev_ms(lambda: lib.vfm_icp_build_system(src2.data_ptr(), tgt.data_ptr(), valid.data_ptr(), n, 1.0, out.data_ptr(), st)), n * 49)
