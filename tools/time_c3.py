"""Timing of the C3 feature stages on one MI355X: ViT-S/14 on 6 x 1200x1600, 6-camera projection +
fused lifting of 20k points.  Prints per-stage milliseconds (HIP events, median of 10)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from vfmreg import ops  # noqa: E402
from vfmreg import vit as V  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


rng = np.random.default_rng(0)
B, H, W, n = 6, 1200, 1600, 20000
imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).cuda()
model = V.ViTS14(V.random_weights(0), H, W)
grids = model.forward(imgs)
t_vit = timed(lambda: model.forward(imgs))
xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 6, n)]
pcl = torch.from_numpy(np.ascontiguousarray(np.insert(xyz, 3, 1, axis=1).T)).cuda()
K = np.array([[800.0, 0, 800], [0, 800, 600], [0, 0, 1]])
Ps = []
for i in range(6):
    y = np.deg2rad(60 * i)
    R = np.stack([[np.sin(y), -np.cos(y), 0], [0, 0, -1], [np.cos(y), np.sin(y), 0]])
    Ps.append(K @ np.c_[R, np.zeros(3)])
desc = torch.zeros((n, 384), dtype=torch.float32, device="cuda")
filled = torch.zeros(n, dtype=torch.uint8, device="cuda")


def lift():
    desc.zero_()   # (the per-camera API writes only the points it claims)
    filled.zero_()
    for c in range(6):
        u, v, idx, cnt = ops.project_pinhole(ops.PROJ_KITTI, pcl, [Ps[c]], None, 1.0, None, None, H, W)
        ops.gather_bilinear(grids[c], H, W, 0, imgs[c], u, v, idx, cnt, desc, filled)


t_lift_loop = timed(lift)


lift_per_camera = lift


plan = ops.LiftPlan([dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, proj_image=None,
                          grid=grids[c], Hup=H, Wup=W, rot_mode=0, raw_image=imgs[c]) for c in range(6)], 384)


def lift():  # projection fused with the gather, all six cameras in one launch (camera records marshalled once)
    plan(pcl, desc, filled)


t_lift = timed(lift)

# registration of the lifted scan against a 200k-point map (C2 solve) and the whole chain back to back
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402
m = 200000
g = torch.Generator(device="cuda").manual_seed(3)
b_desc = torch.randn(m, 384, device="cuda", generator=g)
lift()
pick = torch.randperm(m, device="cuda", generator=g)[:n]
b_desc[pick] = desc + 0.02 * desc.abs().mean() * torch.randn(n, 384, device="cuda", generator=g)
b_xyz = torch.rand(m, 3, device="cuda", generator=g, dtype=torch.float64) * 100.0
q_xyz = torch.from_numpy(np.ascontiguousarray(xyz)).cuda()
b_xyz[pick] = q_xyz + 0.02 * torch.randn(n, 3, device="cuda", generator=g, dtype=torch.float64)
pipe = RegistrationPipeline(n, m, 384, n_iter=50000)
t_reg = timed(lambda: pipe.register(desc, q_xyz, b_desc, b_xyz))


def chain():
    model.forward(imgs, out=grids)
    lift()
    return pipe.register(desc, q_xyz, b_desc, b_xyz)


t_all = timed(chain)
out = chain()
torch.cuda.synchronize()
flops = 6 * 16.6e9
print(f"ViT-S/14 6x{H}x{W}: {t_vit:.3f} ms ({flops / t_vit / 1e9:.1f} TFLOP/s of ~1e11 FLOP)")
print(f"projection + lifting, 6 cameras x {n} points: one launch {t_lift:.3f} ms (per-camera launches: "
      f"{t_lift_loop:.3f} ms); lifted {int(filled.sum())} points")
print(f"registration of the lifted scan vs {m}-point map (50k RANSAC iterations): {t_reg:.3f} ms; "
      f"{int(out['count'].item())} correspondences, |t| = {float(out['T'][:3, 3].norm()):.3f} m")
print(f"C3 end to end, device resident (ViT -> project/lift -> match -> RANSAC), one pair: {t_all:.3f} ms")
