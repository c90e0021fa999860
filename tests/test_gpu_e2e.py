"""BASELINE config C3 end to end on the GPU: DINOv2 ViT-S/14 features of 6 x 1600x1200 surround
images -> LiDAR projection into 6 pinhole cameras -> fused descriptor lifting (first camera wins)
-> descriptor matching against a precomputed map -> RANSAC pose.  Each stage is checked against the
oracle; the solve stages are fed the SAME (GPU-lifted) descriptors so that they compare exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _cameras(n_cam=6, fx=800.0, cx=800.0, cy=600.0):
    """pinhole cameras yawed 60 degrees apart around the LiDAR"""
    K = np.array([[fx, 0, cx], [0, fx, cy], [0, 0, 1.0]])
    Ps = []
    for i in range(n_cam):
        yaw = np.deg2rad(60.0 * i)
        fwd = np.array([np.cos(yaw), np.sin(yaw), 0.0])          # optical axis in the LiDAR frame
        right = np.array([np.sin(yaw), -np.cos(yaw), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        R = np.stack([right, down, fwd])                          # rows: camera x, y, z
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = -R @ np.array([0.1 * np.cos(yaw), 0.1 * np.sin(yaw), 0.3])
        Ps.append(K @ T[:3, :])
    return Ps


def test_c3_end_to_end():
    from oracle import oracle as orc
    from tests.test_gpu_vit import _smooth_images
    from vfmreg import synth
    from vfmreg import vit as V
    from vfmreg.dataloader import KittiOdometry
    from vfmreg.image_features import ImageFeatureGenerator
    from vfmreg.pipeline import RegistrationPipeline
    from vfmreg.prepare_scenes import create_descriptors

    rng = np.random.default_rng(7)
    n, m, H, W = 20000, 200000, 1200, 1600
    cams = [f"cam{i}" for i in range(6)]
    Ps = _cameras()
    imgs = _smooth_images(rng, 6, H, W)
    imgs[2, 100:300, 200:900] = 0                                  # black block: zero descriptors (PS:57-62)
    images = {c: imgs[i] for i, c in enumerate(cams)}

    class Surround:                                                # duck-typed `sequence` (PS:50-76)
        cameras = cams
        image_subsample = 1

        def __init__(self):
            self._k = {c: KittiOdometry({"P2": Ps[i], "Tr_velo_to_cam": np.eye(4)}) for i, c in enumerate(cams)}

        def read_images(self, filenames=None):
            return images

        def project_pcl_to_image(self, pcl, image, camera, _device_inputs=None):
            return self._k[camera].project_pcl_to_image(pcl, image, "camera", _device_inputs=_device_inputs)

        def projection_params(self, camera, image_shape):       # enables the fused multi-camera lift
            return self._k[camera].projection_params("camera", image_shape)

    scan_xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2.5, 6, n)].astype(np.float32)
    w = V.random_weights(seed=0)
    gen = ImageFeatureGenerator("dinov2", use_featup=False, weights=w)
    desc = create_descriptors(None, Surround(), gen, scan_xyz)     # [n, 384] fp32, rows A1+A2+A3 on the GPU
    assert desc.shape == (n, 384) and desc.dtype == np.float32
    # the fused "projection + gather, all cameras in one launch" path must equal the per-camera path bit for bit
    from vfmreg import prepare_scenes as PS
    PS._FORCE_PER_CAMERA = True
    try:
        desc_loop = create_descriptors(None, Surround(), gen, scan_xyz)
    finally:
        PS._FORCE_PER_CAMERA = False
    np.testing.assert_array_equal(desc, desc_loop)

    # ---- stage parity: projection indices exact, lifted values within the ViT tolerance
    grids = orc.vit_reference(w, imgs)
    pcl = np.insert(scan_xyz, 3, values=1, axis=1).T
    ocams = []
    for i in range(6):
        u, v, idx = orc.project(2, pcl, [Ps[i]], None, 1.0, None, None, H, W)
        ocams.append(dict(grid=grids[i], Hup=H, Wup=W, rot_mode=0, black=np.all(imgs[i] == 0, -1), u=u, v=v, idx=idx))
    ref = orc.create_descriptors(n, ocams)
    np.testing.assert_array_equal(np.abs(desc).sum(1) > 0, np.abs(ref).sum(1) > 0)
    seen = np.abs(ref).sum(1) > 0
    assert 0.5 < seen.mean() < 1.0 and np.abs(desc - ref).max() < 1e-2

    # ---- map with precomputed descriptors: the scan's true counterparts + clutter
    T_gt = synth.random_pose(rng)
    pick = rng.permutation(m)[:n]
    b_xyz = np.c_[rng.uniform(-60, 60, m), rng.uniform(-60, 60, m), rng.uniform(-3, 12, m)]
    b_xyz[pick] = scan_xyz.astype(np.float64) @ T_gt[:3, :3].T + T_gt[:3, 3] + rng.normal(0, 0.02, (n, 3))
    b_desc = rng.standard_normal((m, 384)).astype(np.float32)
    b_desc[pick] = desc + 0.02 * np.abs(desc).mean() * rng.standard_normal(desc.shape).astype(np.float32)
    b_desc[pick[~seen]] = rng.standard_normal((int((~seen).sum()), 384)).astype(np.float32)

    pipe = RegistrationPipeline(n, m, 384, n_iter=50000, max_corr_dist=1.0)
    dev = lambda a, t: torch.from_numpy(np.ascontiguousarray(a, dtype=t)).cuda()
    out = pipe.register(dev(desc, np.float32), dev(scan_xyz, np.float64), dev(b_desc, np.float32), dev(b_xyz, np.float64))
    torch.cuda.synchronize()
    T = out["T"].cpu().numpy()
    assert np.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < 0.3

    # ---- solve parity on identical descriptors: indices, inlier mask and pose vs the oracle
    qn, _ = orc.l2norm_rows(desc)
    bn, _ = orc.l2norm_rows(b_desc)
    idx, sim = orc.match_ip_top1(qn, bn)
    got = out["idx"].cpu().numpy()
    solved = got >= 0   # the pipeline leaves queries that provably cannot reach the cosine gate unresolved (-1, -2.0)
    np.testing.assert_array_equal(got[solved], idx[solved])
    assert (sim[~solved] < 0.8).all() and (out["sim"].cpu().numpy()[~solved] == -2.0).all()
    assert (sim[~seen] == 0).all()                                  # zero descriptors never match
    keep = orc.threshold_compact(sim, 0.8)
    k = int(out["count"].item())
    assert k == len(keep) and k > 1000
    corres = np.stack([keep, idx[keep]], 1).astype(np.int32)
    r = orc.ransac_corr(scan_xyz.astype(np.float64), b_xyz, corres, 1.0, 50000, seed=42)
    np.testing.assert_array_equal(out["mask"][:k].cpu().numpy(), r.inlier_mask)
    assert np.linalg.norm(T - r.transformation) <= 1e-5
    np.testing.assert_array_equal(T, r.transformation)


def test_c5_solve_at_full_size():
    """BASELINE config C5 (stretch): 50k-point scan vs 1M-point map, 768-D descriptors, fp16 coarse
    distances + exact decision, RANSAC.  Matching indices are checked against the oracle on every row,
    the solve against the oracle on the GPU's own correspondence set (bit-exact mask and pose), the pose
    against the planted one; the two-stage pipeline must return what the serial one returns."""
    from oracle import oracle as orc
    from vfmreg import synth
    from vfmreg.pipeline import RegistrationPipeline

    n, m, d, iters = 50000, 1000000, 768, 20000
    p = synth.make_pair_device(n, m, d, seed=77)
    outs = []
    # the serial and the overlapped pipeline with `auto` (first registration: best-score int8 records + the half-width probe),
    # then the kernel bench.py's C5 line names, pinned: the half-width pass in fp6 (VFM_RECORDS_MX6_HALF_FUSED = 8,
    # match_coarse_mx6q2_kernel<6, MX6_FUSE, false, 12, 4>) -- every form must return the same bits
    for overlap, coarse in ((False, "auto"), (True, "auto"), (True, "mx6-half")):
        pipe = RegistrationPipeline(n, m, d, n_iter=iters, overlap_ransac=overlap, coarse=coarse)
        if coarse == "mx6-half":
            assert pipe._records() == 8 and pipe.half and pipe.mx6_half
        else:
            assert pipe._records() == 0
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        pipe.synchronize()
        torch.cuda.synchronize()
        outs.append({k: out[k].clone() for k in ("T", "idx", "sim", "count", "mask", "corres", "best_hyp")})
        del pipe
    out = outs[0]
    k = int(out["count"].item())
    assert k > 20000
    for key in outs[0]:  # rows past the correspondence count are never written
        a, b = (outs[0][key][:k], outs[1][key][:k]) if key in ("corres", "mask") else (outs[0][key], outs[1][key])
        assert torch.equal(a, b), key
    # the fp6 half-width pass resolves what it cannot prove below the gate: identical where both resolve, identical matches kept
    h = outs[2]
    for key in ("T", "count", "best_hyp"):
        assert torch.equal(h[key], out[key]), key
    for key in ("corres", "mask"):
        assert torch.equal(h[key][:k], out[key][:k]), key
    both = (h["idx"] >= 0) & (out["idx"] >= 0)
    assert torch.equal(h["idx"][both], out["idx"][both]) and torch.equal(h["sim"][both], out["sim"][both])
    assert int(((h["idx"] >= 0) & (out["idx"] < 0)).sum()) == 0 and bool((out["sim"][(out["idx"] >= 0) & (h["idx"] < 0)] < 0.8).all())
    T = out["T"].cpu().numpy()
    T_gt = p["T_gt"].cpu().numpy() if hasattr(p["T_gt"], "cpu") else np.asarray(p["T_gt"])
    assert np.linalg.norm(T - T_gt) < 0.05
    # matching parity on ALL 50 000 rows (round 4 compared every 100th): BLAS prefilter + exact fp64 decision of the
    # oracle, slabs of rows on a few host threads (numpy and the ctypes call release the GIL: one slab's GEMM runs beside another's scan)
    from concurrent.futures import ThreadPoolExecutor
    qn, _ = orc.l2norm_rows(p["q_desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
    slabs = [(s0, min(s0 + 2048, n)) for s0 in range(0, n, 2048)]
    with ThreadPoolExecutor(max_workers=6) as ex:
        parts = list(ex.map(lambda ab: orc.match_ip_top1(qn[ab[0]:ab[1]], bn, block=512), slabs))
    idx_ref = np.concatenate([a for a, _ in parts])
    sim_ref = np.concatenate([b for _, b in parts])
    assert idx_ref.shape == (n,)
    for which, o in (("auto (best-score int8 records)", out), ("mx6-half (record kind 8)", h)):
        got_i, got_s = o["idx"].cpu().numpy(), o["sim"].cpu().numpy()
        solved = got_i >= 0   # unresolved rows (-1, -2.0): provably below the cosine gate
        assert solved.sum() >= 20000, which
        np.testing.assert_array_equal(got_i[solved], idx_ref[solved], err_msg=which)
        np.testing.assert_array_equal(got_s[solved], sim_ref[solved], err_msg=which)
        assert (sim_ref[~solved] < 0.8).all() and (got_s[~solved] == -2.0).all(), which
    # solve parity on the GPU's correspondences
    corres = out["corres"][:k].cpu().numpy()
    r = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, iters, seed=42)
    assert out["best_hyp"].item() == r.best_hyp
    np.testing.assert_array_equal(T, r.transformation)
    np.testing.assert_array_equal(out["mask"][:k].cpu().numpy(), r.inlier_mask)
    # ... and with the map stored in fp16 (configs[4]: "fp16 descriptor storage"; 1.54 GB instead of 3.07): the registration of the
    # widened rows, bit for bit -- against the same pipeline fed map.half().float(), and against the oracle on every 100th row of it
    del outs, h
    b16 = p["b_desc"].half().contiguous()
    res = []
    for bdesc in (b16, b16.float().contiguous()):
        pipe = RegistrationPipeline(n, m, d, n_iter=iters, overlap_ransac=True, coarse="mx6-half")
        o = pipe.register(p["q_desc"], p["q_xyz"], bdesc, p["b_xyz"])
        pipe.synchronize()
        torch.cuda.synchronize()
        res.append({kk: o[kk].clone() for kk in ("T", "idx", "sim", "count", "mask", "corres", "best_hyp")})
        del pipe, o
    k16 = int(res[1]["count"].item())
    assert k16 > 20000 and int(res[0]["count"].item()) == k16
    for key in ("T", "best_hyp", "idx", "sim"):
        assert torch.equal(res[0][key], res[1][key]), key
    for key in ("corres", "mask"):
        assert torch.equal(res[0][key][:k16], res[1][key][:k16]), key
    assert np.linalg.norm(res[0]["T"].cpu().numpy() - T_gt) < 0.05
    rows = torch.arange(0, n, 100, device="cuda")
    qn16, _ = orc.l2norm_rows(p["q_desc"][rows].cpu().numpy())
    bn16, _ = orc.l2norm_rows(b16.float().cpu().numpy())
    idx16, sim16 = orc.match_ip_top1(qn16, bn16)
    gi, gs = res[0]["idx"][rows].cpu().numpy(), res[0]["sim"][rows].cpu().numpy()
    solved = gi >= 0
    assert solved.sum() >= 50
    np.testing.assert_array_equal(gi[solved], idx16[solved])
    np.testing.assert_array_equal(gs[solved], sim16[solved])
    assert (sim16[~solved] < 0.8).all()


def test_c3_pipelined_equals_the_stage_by_stage_path_and_the_oracle():
    """EndToEndPipeline (round 4): the feature stage of pair i + 1 (ViT + projection / lifting on a stream of its own) beside the
    registration of pair i.  For a sequence of different pairs every output -- lifted descriptors, correspondences, inlier mask,
    pose, winner -- is bit-equal to model.forward -> LiftPlan -> RegistrationPipeline.register run one after the other, and the
    solve of a registration equals the oracle's on the same descriptors (as test_c3_end_to_end)."""
    from oracle import oracle as orc
    from tests.test_gpu_vit import _smooth_images
    from vfmreg import ops
    from vfmreg import vit as V
    from vfmreg.pipeline import EndToEndPipeline, RegistrationPipeline

    rng = np.random.default_rng(31)
    n, m, H, W, npairs = 6000, 40000, 560, 700, 7
    Ps = _cameras()
    for P in Ps:
        P[0] *= W / 1600.0
        P[1] *= H / 1200.0
    cams = [dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, rot_mode=0) for c in range(6)]
    model = V.ViTS14(V.random_weights(seed=4, dim=384, depth=3, mlp=1536), H, W)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()   # noqa: E731
    pairs = []
    for p in range(npairs):
        imgs = dev(_smooth_images(np.random.default_rng(100 + p), 6, H, W), np.uint8)
        xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2.5, 6, n)]
        pairs.append(dict(imgs=imgs, pcl=dev(np.insert(xyz, 3, 1, axis=1).T, np.float64), q_xyz=dev(xyz, np.float64)))
    # maps: the scan's own lifted descriptors (stage by stage) + noise, planted pose
    ref = []
    reg = RegistrationPipeline(n, m, 384, n_iter=5000)
    g = torch.Generator(device="cuda").manual_seed(5)
    for p in pairs:
        grids = model.forward(p["imgs"]).clone()
        desc = torch.empty((n, 384), dtype=torch.float32, device="cuda")
        filled = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ops.LiftPlan([dict(c, proj_image=None, grid=grids[k], Hup=H, Wup=W, raw_image=p["imgs"][k]) for k, c in enumerate(cams)], 384)(p["pcl"], desc, filled)
        b_desc = torch.randn(m, 384, device="cuda", generator=g)
        pick = torch.randperm(m, device="cuda", generator=g)[:n]
        b_desc[pick] = desc + 0.05 * desc.abs().mean() * torch.randn(n, 384, device="cuda", generator=g)
        b_xyz = torch.rand(m, 3, device="cuda", generator=g, dtype=torch.float64) * 100.0
        b_xyz[pick] = p["q_xyz"] + 0.02 * torch.randn(n, 3, device="cuda", generator=g, dtype=torch.float64)
        p.update(b_desc=b_desc.contiguous(), b_xyz=b_xyz.contiguous())
        out = reg.register(desc, p["q_xyz"], p["b_desc"], p["b_xyz"])
        torch.cuda.synchronize()
        ref.append({k: out[k].clone() for k in ("T", "count", "corres", "mask", "best_hyp", "idx", "sim")} | {"desc": desc.clone()})
    e2e = EndToEndPipeline(model, cams, n, m, n_iter=5000, depth=4)
    snaps = []
    for p in pairs:                       # back to back: no host synchronisation between the pairs
        out = e2e.submit(p["imgs"], p["pcl"], p["q_xyz"], p["b_desc"], p["b_xyz"])
        with torch.cuda.stream(out["result_stream"]):
            snaps.append({k: out[k].clone() for k in ("T", "count", "corres", "mask", "best_hyp", "desc")})
    e2e.synchronize()
    torch.cuda.synchronize()
    for i, (s, r) in enumerate(zip(snaps, ref)):
        assert torch.equal(s["desc"], r["desc"]), i
        c = int(r["count"].item())
        assert c > 1000 and int(s["count"].item()) == c, i
        assert torch.equal(s["T"], r["T"]) and torch.equal(s["best_hyp"], r["best_hyp"]), i
        assert torch.equal(s["corres"][:c], r["corres"][:c]) and torch.equal(s["mask"][:c], r["mask"][:c]), i
    # the same pairs in groups of three (3 + 3 + 1): one ViT call per group (EndToEndPipeline.submit_group), everything else as above
    del e2e
    e2g = EndToEndPipeline(model, cams, n, m, n_iter=5000, depth=4, group=3, group_depth=2)
    gsnaps = []

    def keep(k, out):
        with torch.cuda.stream(out["result_stream"]):
            gsnaps.append({k2: out[k2].clone() for k2 in ("T", "count", "corres", "mask", "best_hyp", "desc")})
    for lo in range(0, npairs, 3):
        outs = e2g.submit_group([(p["imgs"], p["pcl"], p["q_xyz"], p["b_desc"], p["b_xyz"]) for p in pairs[lo:lo + 3]], on_result=keep)
        assert len(outs) == min(3, npairs - lo)
    e2g.synchronize()
    torch.cuda.synchronize()
    assert len(gsnaps) == npairs
    for i, (s, r) in enumerate(zip(gsnaps, ref)):
        c = int(r["count"].item())
        assert torch.equal(s["desc"], r["desc"]) and int(s["count"].item()) == c, i
        assert torch.equal(s["T"], r["T"]) and torch.equal(s["best_hyp"], r["best_hyp"]), i
        assert torch.equal(s["corres"][:c], r["corres"][:c]) and torch.equal(s["mask"][:c], r["mask"][:c]), i
    with pytest.raises(ValueError):
        e2g.submit_group([])
    del e2g
    # the oracle's solve on the GPU-lifted descriptors of one pair (the chain's precision is the subject of test_c3_fp16_vit_...)
    p, r = pairs[3], ref[3]
    qn, _ = orc.l2norm_rows(r["desc"].cpu().numpy())
    bn, _ = orc.l2norm_rows(p["b_desc"].cpu().numpy())
    idx, sim = orc.match_ip_top1(qn, bn)
    keep = orc.threshold_compact(sim, 0.8)
    corres = np.stack([keep, idx[keep]], 1).astype(np.int32)
    o = orc.ransac_corr(p["q_xyz"].cpu().numpy(), p["b_xyz"].cpu().numpy(), corres, 10000.0, 5000, seed=42)
    c = int(snaps[3]["count"].item())
    np.testing.assert_array_equal(snaps[3]["corres"][:c].cpu().numpy(), corres)
    np.testing.assert_array_equal(snaps[3]["T"].cpu().numpy(), o.transformation)
    np.testing.assert_array_equal(snaps[3]["mask"][:c].cpu().numpy(), o.inlier_mask)


def test_c3_pipeline_on_compute_unit_masked_streams():
    """EndToEndPipeline(feature_cus > 0): the feature stage and the registration on streams restricted to disjoint sets of compute
    units (hipExtStreamCreateWithCUMask; ADVICE r4: the mask's word count / bit index, the call's argument types) -- same bits as the
    unmasked pipeline."""
    from tests.test_gpu_vit import _smooth_images
    from vfmreg import ops
    from vfmreg import vit as V
    from vfmreg.pipeline import EndToEndPipeline

    rng = np.random.default_rng(41)
    n, m, H, W, npairs = 3000, 20000, 560, 700, 3
    Ps = _cameras()
    for P in Ps:
        P[0] *= W / 1600.0
        P[1] *= H / 1200.0
    cams = [dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, rot_mode=0) for c in range(6)]
    model = V.ViTS14(V.random_weights(seed=6, dim=384, depth=2, mlp=1536), H, W)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()   # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(9)
    pairs = []
    for p in range(npairs):
        imgs = dev(_smooth_images(np.random.default_rng(200 + p), 6, H, W), np.uint8)
        xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2.5, 6, n)]
        pcl, q_xyz = dev(np.insert(xyz, 3, 1, axis=1).T, np.float64), dev(xyz, np.float64)
        grids = model.forward(imgs).clone()
        desc = torch.empty((n, 384), dtype=torch.float32, device="cuda")
        filled = torch.zeros(n, dtype=torch.uint8, device="cuda")
        ops.LiftPlan([dict(c, proj_image=None, grid=grids[k], Hup=H, Wup=W, raw_image=imgs[k]) for k, c in enumerate(cams)], 384)(pcl, desc, filled)
        b_desc = torch.randn(m, 384, device="cuda", generator=g)
        pick = torch.randperm(m, device="cuda", generator=g)[:n]
        b_desc[pick] = desc + 0.05 * desc.abs().mean() * torch.randn(n, 384, device="cuda", generator=g)
        b_xyz = torch.rand(m, 3, device="cuda", generator=g, dtype=torch.float64) * 100.0
        b_xyz[pick] = q_xyz + 0.02 * torch.randn(n, 3, device="cuda", generator=g, dtype=torch.float64)
        pairs.append((imgs, pcl, q_xyz, b_desc.contiguous(), b_xyz.contiguous()))
    res = {}
    for cus in (0, 32):
        e2e = EndToEndPipeline(model, cams, n, m, n_iter=3000, depth=3, feature_cus=cus)
        snaps = []
        for p in pairs:
            out = e2e.submit(*p)
            with torch.cuda.stream(out["result_stream"]):
                snaps.append({k: out[k].clone() for k in ("T", "count", "corres", "mask", "best_hyp", "desc")})
        e2e.synchronize()
        torch.cuda.synchronize()
        res[cus] = snaps
        del e2e
    for a, b in zip(res[0], res[32]):
        c = int(a["count"].item())
        assert c > 500 and int(b["count"].item()) == c
        assert torch.equal(a["desc"], b["desc"]) and torch.equal(a["T"], b["T"]) and torch.equal(a["best_hyp"], b["best_hyp"])
        assert torch.equal(a["corres"][:c], b["corres"][:c]) and torch.equal(a["mask"][:c], b["mask"][:c])
