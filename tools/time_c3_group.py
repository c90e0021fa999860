"""C3 as a pipeline with the feature stages of G pairs sharing one ViT call (EndToEndPipeline.submit_group): registrations per
second from uint8 images against G (1 = EndToEndPipeline.submit, one ViT call per pair).  Output: profiles/r04_time_c3_group.txt."""
import sys, time, gc
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import ops  # noqa: E402
from vfmreg import vit as V  # noqa: E402
from vfmreg.pipeline import EndToEndPipeline  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
B, H, W, n, m = 6, 1200, 1600, 20000, 200000
imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).to(dev)
model = V.ViTS14(V.random_weights(0), H, W, device=dev)
xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 6, n)]
pcl = torch.from_numpy(np.ascontiguousarray(np.insert(xyz, 3, 1, axis=1).T)).to(dev)
K = np.array([[800.0, 0, 800], [0, 800, 600], [0, 0, 1]])
Ps = []
for i in range(6):
    y = np.deg2rad(60 * i)
    R = np.stack([[np.sin(y), -np.cos(y), 0], [0, 0, -1], [np.cos(y), np.sin(y), 0]])
    Ps.append(K @ np.c_[R, np.zeros(3)])
rig = [dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, rot_mode=0) for c in range(6)]
grids = model.forward(imgs)
desc = torch.empty((n, 384), dtype=torch.float32, device=dev)
filled = torch.zeros(n, dtype=torch.uint8, device=dev)
ops.LiftPlan([dict(c, proj_image=None, grid=grids[k], Hup=H, Wup=W, raw_image=imgs[k]) for k, c in enumerate(rig)], 384)(pcl, desc, filled)
g = torch.Generator(device=dev).manual_seed(3)
b_desc = torch.randn(m, 384, device=dev, generator=g)
pick = torch.randperm(m, device=dev, generator=g)[:n]
b_desc[pick] = desc + 0.02 * desc.abs().mean() * torch.randn(n, 384, device=dev, generator=g)
b_xyz = torch.rand(m, 3, device=dev, generator=g, dtype=torch.float64) * 100.0
q_xyz = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev)
b_xyz[pick] = q_xyz + 0.02 * torch.randn(n, 3, device=dev, generator=g, dtype=torch.float64)
img_sets = [imgs, torch.flip(imgs, dims=[0]).contiguous()]
STEPS = 192
import os
from vfmreg import pipeline as _pl
_pl.FEATURE_QUEUE_SKIP = int(os.environ.get("VFM_FEATURE_SKIP", _pl.FEATURE_QUEUE_SKIP))
_pl.E2E_SOLVE_STREAMS = int(os.environ.get("VFM_E2E_SOLVE", _pl.E2E_SOLVE_STREAMS))
if os.environ.get("VFM_PREAMBLE"):   # what bench.py has done by the time it measures extra.C3_pipelined: C2 pipelines on D.2 data, then freed
    from vfmreg import synth
    from vfmreg.pipeline import RegistrationPipeline
    prs = [synth.make_pair_device(20000, 200000, 384, seed=42 + j, device=dev) for j in range(4)]
    for mode in os.environ["VFM_PREAMBLE"].split(","):
        pp = RegistrationPipeline(20000, 200000, 384, n_iter=50000, device=dev, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=mode)
        for i in range(200):
            q = prs[i % 4]
            pp.register(q["q_desc"], q["q_xyz"], q["b_desc"], q["b_xyz"])
            pp._poll_feedback()
        pp.synchronize()
        torch.cuda.synchronize()
        del pp
    del prs
    torch.cuda.empty_cache()
for G in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8", "1"])]:
    e2e = EndToEndPipeline(model, rig, n, m, n_iter=50000, depth=4, group=G, group_depth=3)
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    T_even = []   # the pose of the last pair whose cameras are the ones the map's descriptors were lifted from (img_sets[0]: the planted
                  # transform is the identity; img_sets[1] shows the cameras in another order, its pose is not meaningful)

    def snap(out):
        with torch.cuda.stream(out["result_stream"]):
            T_even[:] = [out["T"].clone()]
    for steps in (8, STEPS, STEPS):
        gc.collect(); gc.disable()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if G == 1:
            for i in range(steps):
                res = e2e.submit(img_sets[i % 2], pcl, q_xyz, b_desc, b_xyz, inputs_ready=ready)
                if i % 2 == 0:
                    snap(res)
                e2e.reg._poll_feedback()
        else:
            for lo in range(0, steps, G):
                res = e2e.submit_group([(img_sets[i % 2], pcl, q_xyz, b_desc, b_xyz) for i in range(lo, min(lo + G, steps))],
                                       inputs_ready=ready, on_result=lambda k, out, lo=lo: snap(out) if (lo + k) % 2 == 0 else None)[-1]
                e2e.reg._poll_feedback()
        e2e.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
    print(f"{G} pair(s) per ViT call: {steps / dt:7.1f} registrations/s ({1e3 * dt / steps:.3f} ms per pair), records {e2e.reg._records()}, "
          f"correspondences {int(res['count'].item())}, pose error vs planted {float((T_even[0].cpu() - torch.eye(4, dtype=torch.float64)).norm()):.1e}", flush=True)
    del e2e
