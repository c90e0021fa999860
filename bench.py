#!/usr/bin/env python
"""bench.py -- registrations/sec of the correspondence-and-solve hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1] ("C2"): one step = one registration of a 20 000-point scan
against a 200 000-point map with 384-D descriptors (precomputed, resident in HBM) and 50 000 RANSAC
iterations: normalise + int8 fragment conversion of BOTH clouds (the reference renormalises the map
on every call, VoxelHashMap.cpp:469-482), exact top-1 inner-product search (int8 MFMA coarse pass with
proven bounds -> fp32 refinement -> fp64 decision: the oracle's indices), cosine >= 0.8 threshold
and compaction, correspondence RANSAC with 3-point Kabsch.  Synthetic inputs of SURVEY.md 8 D.2.

Multi-GPU: independent scene pairs are sharded across ranks, no data-path
collective; one all_gather of the 4x4 poses (RCCL) closes the timed region.  Weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- the dominant kernel (int8 MFMA coarse pass): algorithmic operations 2*N*M*D per launch
                  / average launch duration measured with HIP events on its stream, vs the dense int8
                  MFMA peak (5 POP/s = 2 x the 2.5 PFLOP/s fp16 figure of MI355X_MICROARCH.md: the
                  32x32x32 i8 instruction has twice the k of 32x32x16 f16 at the same issue rate);
  cpu_baseline -- the CPU oracle (oracle/, a port: faiss and Open3D are absent) timed on this
                  box's host cores on a bounded sample of the same workload;
  extra        -- outside the timed region: |T_gpu - T_oracle|_F of one pair ("pose delta vs ref"); C2_full_width / C2_sustained /
                  C2_lifted (the same timed-loop form with the full-width pass, over 200 steps, and on descriptors that look like
                  lifted ViT features -- figures that do not rest on D.2's prunable noise); config C3 end to end (+ the ViT's
                  roofline entry) and config C5 (+ its coarse kernel's roofline entry).
--pairs P (config C4): P independent scene pairs, pair p generated from seed 42 + p on rank p mod N and registered
there; every rank prints its own rate to stderr before the gather.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "vfm-registration_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

N_SCAN, N_MAP, DIM, RANSAC_ITERS = 20000, 200000, 384, 50000
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/fp16 MFMA ~2.5 PFLOP/s
MFMA_I8_PEAK_TOPS = 5000.0     # dense int8 MFMA: 2 x the fp16 rate (same table: "I8 ~2x bf16 rate (2xK)", ubench >= 4404)
SETTLE = 6   # registrations the auto policy gets, one at a time, before the warm-up (see main)
MFMA_F6_PEAK_TFLOPS = 10000.0  # dense fp6 / fp4 scaled MFMA (same table: "~10 PF dense", FP6 ubench >= 7287; tools/probe/mx6_probe.hip: 6400)


C3_GROUP = 7   # pairs whose cameras share one ViT call in the grouped C3 pipeline (tools/time_c3_group.py; 42 images: one full round of the
               # fused QKV + attention kernel's workgroups -- round 5: 4)
C3_GROUP_ALSO = (4,)   # ... and reported beside it


def cpu_baseline(p, iters=RANSAC_ITERS, T_gpu=None):
    """Reference-CPU-path stand-in (kind 'port'): the oracle's restatement -- fp32 BLAS Q.B^T + exact
    fp64 decision, threshold, OpenMP RANSAC -- on the SAME scene pair the GPU registered (host copies `p`),
    on a bounded sample, extrapolated linearly.  When the whole registration fits the time budget it is run
    completely, which also yields the metric's "pose delta vs ref": |T_gpu - T_oracle|_F."""
    import numpy as np
    from oracle import oracle as orc

    n, d = p["q_desc"].shape
    m = p["b_desc"].shape[0]
    # 1) probe on a small sample to size the run: the whole registration is timed when it fits in
    #    ~40 s of host time, otherwise a bounded sample is extrapolated linearly (stated in `sample`).
    rows = 512
    t0 = time.perf_counter()
    bn, _ = orc.l2norm_rows(p["b_desc"])
    t_norm_map = time.perf_counter() - t0
    qn, _ = orc.l2norm_rows(p["q_desc"])
    t0 = time.perf_counter()
    orc.match_ip_top1(qn[:rows], bn)
    t_probe = time.perf_counter() - t0
    full = t_probe * (n / rows) < 40.0
    rows_used = n if full else 4 * rows
    t0 = time.perf_counter()
    idx, sim = orc.match_ip_top1(qn[:rows_used], bn)
    t_match = time.perf_counter() - t0
    keep = orc.threshold_compact(sim, 0.8)
    corres = np.stack([keep, idx[keep]], 1).astype(np.int32)
    if not full:  # RANSAC cost ~ iters * C: bring C to the full workload's by tiling the sample
        reps = max(1, int(round((n / rows_used))))
        corres = np.tile(corres, (reps, 1))
    it_used = iters if full else max(1000, iters // 10)
    t0 = time.perf_counter()
    res = orc.ransac_corr(p["q_xyz"], p["b_xyz"], corres, 10000.0, it_used, seed=42)
    t_ransac = time.perf_counter() - t0
    total = t_norm_map + t_match * (n / rows_used) + t_ransac * (iters / it_used)
    what = "the WHOLE registration (no extrapolation)" if full else "a bounded sample, extrapolated linearly"
    out = {
        "value": 1.0 / total, "unit": "registrations/s", "cores": orc.num_threads(), "kind": "port",
        "sample": (f"CPU oracle (numpy BLAS fp32 Q.B^T prefilter + C/OpenMP fp64 decision and RANSAC) on {what}: "
                   f"map renorm {m}x{d} {t_norm_map:.2f}s, search of {rows_used}/{n} scan rows vs the full map "
                   f"{t_match:.2f}s, RANSAC {it_used}/{iters} iterations over {len(corres)} correspondences "
                   f"{t_ransac:.2f}s -> {total:.1f}s per registration; pose err vs planted "
                   f"{float(np.linalg.norm(res.transformation - p['T_gt'])):.4f}"),
        "host_cpu_count": os.cpu_count(),
    }
    delta = None
    if full and T_gpu is not None:
        delta = {"pose_delta_vs_oracle_frobenius": float(np.linalg.norm(np.asarray(T_gpu) - res.transformation)),
                 "correspondences_oracle": int(len(keep)),
                 "note": "GPU pose of global pair 0 vs the CPU oracle's pose on identical inputs (same seed, same RANSAC stream)"}
    return out, delta


def live_traffic(records_kind: int, timeout_s: int = 60):
    """HBM bytes per launch of the coarse kernel, measured while bench.py runs: FETCH_SIZE and WRITE_SIZE from two SEPARATE
    `rocprofv3 --kernel-trace --pmc <counter>` passes (MI355X_MICROARCH.md, HBM / rocprofv3 section: no other trace domain beside
    --pmc) of tools/prof_match.py -- the matching stage of config C2 at the record kind the timed region ran, three launches --
    each in a subprocess of its own.  Units and the gfx950 correction as tools/pmc_coarse.sh applies them (both counters in KB;
    FETCH_SIZE doubled).  Returns None when rocprofv3 is missing, fails or reports no such kernel: the line then carries the
    round's committed passes and says so."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    t0 = time.perf_counter()
    got, kernel = {}, ""
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                env = dict(os.environ, VFM_RECORDS=str(records_kind), TMPDIR="/tmp")
                subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", td, "-o", ctr, "--",
                                sys.executable, str(ROOT / "tools" / "prof_match.py"), "3"],
                               cwd="/tmp", env=env, capture_output=True, timeout=timeout_s, check=True)
                vals = []
                for f in glob.glob(f"{td}/**/{ctr}_counter_collection.csv", recursive=True):
                    for r in csv.DictReader(open(f)):
                        if "match_coarse" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                            kernel = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
                            vals.append(float(r["Counter_Value"]))
                if not vals:
                    return None
                got[ctr] = vals
    except Exception as e:  # never lose the line to the profiler
        print(f"[bench] live traffic measurement failed ({type(e).__name__}: {e}); using the committed passes", file=sys.stderr, flush=True)
        return None
    fetch, write = (sum(got[c]) / len(got[c]) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    return {"hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
            "launches": len(got["FETCH_SIZE"]), "kernel": kernel, "seconds": time.perf_counter() - t0}


def extra_configs(dev):
    """Other BASELINE configs, measured OUTSIDE the timed region (information only; the headline stays C2):
    C3 = C2 + DINOv2 ViT-S/14 on 6 x 1200 x 1600 + 6-camera projection/lifting, one pair end to end (latency);
    C5 = 50k x 1M x 768 (stretch)."""
    import numpy as np
    import torch
    from vfmreg import _lib, ops, synth
    from vfmreg import vit as V
    from vfmreg.pipeline import RegistrationPipeline
    lib = _lib.load()

    def timed(fn, reps=7):
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

    out = {}
    # ---- C3
    rng = np.random.default_rng(0)
    B, H, W, n, m = 6, 1200, 1600, N_SCAN, N_MAP
    imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).to(dev)
    model = V.ViTS14(V.random_weights(0), H, W, device=dev)
    grids = model.forward(imgs)
    t_vit = timed(lambda: model.forward(imgs))
    xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 6, n)]
    pcl = torch.from_numpy(np.ascontiguousarray(np.insert(xyz, 3, 1, axis=1).T)).to(dev)
    K = np.array([[800.0, 0, 800], [0, 800, 600], [0, 0, 1]])
    Ps = []
    for i in range(6):
        y = np.deg2rad(60 * i)
        R = np.stack([[np.sin(y), -np.cos(y), 0], [0, 0, -1], [np.cos(y), np.sin(y), 0]])
        Ps.append(K @ np.c_[R, np.zeros(3)])
    desc = torch.empty((n, 384), dtype=torch.float32, device=dev)   # (every row is written by the kernel: no clearing pass)
    filled = torch.zeros(n, dtype=torch.uint8, device=dev)

    # (the camera records are marshalled once: ops.LiftPlan; the ViT writes its patch grids into the same buffer every call)
    plan = ops.LiftPlan([dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, proj_image=None,
                              grid=grids[c], Hup=H, Wup=W, rot_mode=0, raw_image=imgs[c]) for c in range(6)], 384)

    def lift():
        plan(pcl, desc, filled)
    t_lift = timed(lift)
    g = torch.Generator(device=dev).manual_seed(3)
    b_desc = torch.randn(m, 384, device=dev, generator=g)
    pick = torch.randperm(m, device=dev, generator=g)[:n]
    b_desc[pick] = desc + 0.02 * desc.abs().mean() * torch.randn(n, 384, device=dev, generator=g)
    b_xyz = torch.rand(m, 3, device=dev, generator=g, dtype=torch.float64) * 100.0
    q_xyz = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev)
    b_xyz[pick] = q_xyz + 0.02 * torch.randn(n, 3, device=dev, generator=g, dtype=torch.float64)
    pipe = RegistrationPipeline(n, m, 384, n_iter=RANSAC_ITERS, device=dev)
    t_reg = timed(lambda: pipe.register(desc, q_xyz, b_desc, b_xyz))

    def chain():
        model.forward(imgs, out=grids)
        lift()
        return pipe.register(desc, q_xyz, b_desc, b_xyz)
    t_all = timed(chain)
    r = chain()
    torch.cuda.synchronize()
    vit_flops = 6 * 16.6e9  # 16.6 GFLOP per 224 x 294 image
    out["C3"] = {"workload": "one pair end to end, device resident: ViT-S/14 on 6 x 1200x1600 -> 6-camera projection + lifting "
                             "of 20000 points -> match vs 200000-point map -> 50000-iteration RANSAC (latency, no pipelining)",
                 "ms_end_to_end": t_all, "ms_vit": t_vit, "ms_project_lift": t_lift, "ms_registration": t_reg,
                 "correspondences": int(r["count"].item()),
                 "vit_roofline": {"bound": "mfma", "flops": vit_flops, "achieved": vit_flops / (t_vit * 1e-3) / 1e12,
                                  "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": vit_flops / (t_vit * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS}}
    # ---- C3 as a pipeline: ViT + lifting of pair i + 1 on a stream of its own beside the registration of pair i
    # (vfmreg.pipeline.EndToEndPipeline; parity: tests/test_gpu_e2e.py::test_c3_pipelined_...).  Throughput from uint8 images.
    try:
        import gc
        import time as _time
        from vfmreg.pipeline import EndToEndPipeline
        del pipe
        rig = [dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, rot_mode=0) for c in range(6)]
        e2e = EndToEndPipeline(model, rig, n, m, n_iter=RANSAC_ITERS, depth=4, device=dev)
        img_sets = [imgs, torch.flip(imgs, dims=[0]).contiguous()]   # two different surround views of the same rig
        torch.cuda.synchronize()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        res = None
        for steps_e2e in (8, 60):                                     # settle the policy / warm up, then the timed run
            gc.collect()
            gc.disable()
            torch.cuda.synchronize()
            t0 = _time.perf_counter()
            for i in range(steps_e2e):
                res = e2e.submit(img_sets[i % 2], pcl, q_xyz, b_desc, b_xyz, inputs_ready=ready)
                if steps_e2e == 8:        # the settle pass: one pair at a time, so that the policy reads every search's feedback (as --form c3 does)
                    e2e.synchronize()
                    torch.cuda.synchronize()
                e2e.reg._poll_feedback()
            e2e.synchronize()
            torch.cuda.synchronize()
            dt = _time.perf_counter() - t0
            gc.enable()
        out["C3_pipelined"] = {"workload": "C3 end to end as a pipeline over independent pairs, from uint8 images: ViT-S/14 on 6 x 1200x1600 + "
                                           "6-camera projection / lifting of pair i + 1 on a stream of its own beside the registration of pair i "
                                           "(20000 lifted points vs 200000-point map, 50000 RANSAC iterations)",
                               "value": steps_e2e / dt, "unit": "registrations/s", "steps": steps_e2e, "ms_per_step": 1e3 * dt / steps_e2e,
                               "coarse_pass": pass_name(e2e.reg), "correspondences": int(res["count"].item()),
                               "serial_equivalent_ms": t_all}
        del e2e
        # the same job with the feature stages of G pairs sharing one ViT call (EndToEndPipeline.submit_group; parity: the same test)
        for G in (C3_GROUP,) + C3_GROUP_ALSO:
            e2e = EndToEndPipeline(model, rig, n, m, n_iter=RANSAC_ITERS, depth=4, device=dev, group=G, group_depth=3)
            for steps_e2e in (G if G > 8 else 8, 9 * G if 9 * G > 64 else 64):
                gc.collect()
                gc.disable()
                torch.cuda.synchronize()
                t0 = _time.perf_counter()
                for lo in range(0, steps_e2e, G):
                    res = e2e.submit_group([(img_sets[i % 2], pcl, q_xyz, b_desc, b_xyz) for i in range(lo, min(lo + G, steps_e2e))],
                                           inputs_ready=ready)[-1]
                    if steps_e2e <= 8 or steps_e2e == G:    # the settle pass, one group at a time
                        e2e.synchronize()
                        torch.cuda.synchronize()
                    e2e.reg._poll_feedback()
                e2e.synchronize()
                torch.cuda.synchronize()
                dt = _time.perf_counter() - t0
                gc.enable()
            row = {"workload": f"the same pairs, the cameras of {G} pairs per ViT call ({6 * G} images), each pair lifted and registered on its own as before",
                   "pairs_per_vit_call": G, "value": steps_e2e / dt, "unit": "registrations/s", "steps": steps_e2e,
                   "ms_per_step": 1e3 * dt / steps_e2e, "coarse_pass": pass_name(e2e.reg), "correspondences": int(res["count"].item())}
            if G == C3_GROUP:
                out["C3_pipelined"]["grouped"] = row
            else:
                out["C3_pipelined"]["grouped"].setdefault("other_group_sizes", []).append(row)
            del e2e
    except Exception as e:  # never lose the line to an auxiliary measurement
        out["C3_pipelined"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- the ViT on a batch of scans (prepare_scenes.py walks ~170 clouds of a scene: create_descriptors_batch, 28 clouds per call; round 5: 15)
    try:
        for clouds, key in ((28, "ViT_batched"), (14, "at_84_images"), (15, "at_90_images"), (8, "at_48_images")):
            big = imgs.repeat(clouds, 1, 1, 1)
            tb = timed(lambda: model.forward(big), reps=5)
            row = {"workload": f"ViT-S/14 on {6 * clouds} x 1200x1600 images per call ({clouds} clouds of a scene x 6 cameras: prepare_scenes."
                               "create_descriptors_batch's default is 28); LDS-tiled 128 x 128 GEMMs, QKV + attention per (image, head) and fc1 -> GELU -> fc2 "
                               "per 128 tokens in one workgroup each where vfm_vit_forward's policies take them, the token-stationary kernel for fc1 (and QKV otherwise) where "
                               "its rounds of one workgroup per compute unit are full",
                   "images": 6 * clouds, "ms": tb, "ms_per_scan_of_6": tb / clouds,
                   "roofline": {"bound": "mfma", "flops": clouds * vit_flops, "achieved": clouds * vit_flops / (tb * 1e-3) / 1e12,
                                "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": clouds * vit_flops / (tb * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS}}
            if key == "ViT_batched":
                out[key] = row
            else:
                out["ViT_batched"][key] = row
            del big
    except Exception as e:
        out["ViT_batched"] = {"error": f"{type(e).__name__}: {e}"}
    del model, imgs, b_desc
    # ---- the reference-shaped call (numpy in, numpy out): RegistrationNode.ransac_registration(voxel_map, raw_scan, 'vfm') with the
    # scene's map kept between scans (registration_node.py:556-589), with and without the ICP refinement
    try:
        import time as _time
        from vfmreg.mapping import VoxelHashMap
        from vfmreg.registration import RegistrationNode
        VoxelHashMap.quiet = True
        pp = synth.make_pair(N_SCAN, N_MAP, DIM, seed=11)
        voxel_map = np.c_[pp["b_xyz"], pp["b_desc"]].astype(np.float32)
        raw_scan = np.c_[pp["q_xyz"], pp["q_desc"]].astype(np.float32)
        def time_api(vmap, scan):
            api = {}
            for name, icp in (("ms_without_icp", False), ("ms_with_icp", True)):
                node = RegistrationNode(cache_map=True)
                for _ in range(2):
                    node.ransac_registration(vmap, scan, "vfm", run_icp=icp)
                ts = []
                for _ in range(15):
                    torch.cuda.synchronize()
                    t0 = _time.perf_counter()
                    poses = node.ransac_registration(vmap, scan, "vfm", run_icp=icp)
                    torch.cuda.synchronize()
                    ts.append(1e3 * (_time.perf_counter() - t0))
                api[name] = sorted(ts)[len(ts) // 2]
            return api, poses

        api, poses = time_api(voxel_map, raw_scan)
        api["pose_err_vs_planted"] = float(np.linalg.norm(poses[1] - pp["T_gt"]))
        # the same call the two other ways -- (cold) the DEFAULT node, cache_map=False: the map rebuilt from the array in
        # every call as registration_node.py:402-403 does (upload of the 200000 x 387 rows, voxel cap, container replay, cast, search
        # operand), and (handle) RegistrationNode.set_map(): the scene's map built once, explicitly, no fingerprint
        try:
            def med(fn, reps):
                ts = []
                for _ in range(reps):
                    torch.cuda.synchronize()
                    t0 = _time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    ts.append(1e3 * (_time.perf_counter() - t0))
                return sorted(ts)[len(ts) // 2], r
            cold_node = RegistrationNode()
            cold_node.ransac_registration(voxel_map, raw_scan, "vfm")
            api["ms_cold"], pc = med(lambda: cold_node.ransac_registration(voxel_map, raw_scan, "vfm"), 5)
            hnode = RegistrationNode()
            t0 = _time.perf_counter()
            handle = hnode.set_map(voxel_map)
            torch.cuda.synchronize()
            api["ms_set_map"] = 1e3 * (_time.perf_counter() - t0)
            for _ in range(2):
                hnode.ransac_registration(handle, raw_scan, "vfm")
            api["ms_warm_handle"], ph = med(lambda: hnode.ransac_registration(handle, raw_scan, "vfm"), 15)
            api["cold_and_handle_poses_equal_the_warm_pose"] = bool(np.array_equal(pc[0], poses[0]) and np.array_equal(ph[0], poses[0]))
            api["note"] = ("ms_without_icp / ms_with_icp: cache_map=True (opt-in: the map kept while the caller passes the same array); ms_cold: the default "
                           "node, the reference's behaviour (RN:402-403: map rebuilt per call); ms_warm_handle: set_map() once (ms_set_map), then the handle")
            del cold_node, hnode, handle
        except Exception as e:
            api["cold_error"] = f"{type(e).__name__}: {e}"
        out["API_ransac_registration"] = dict(api, workload="RegistrationNode.ransac_registration(voxel_map, raw_scan, 'vfm'): numpy in / numpy out, "
                                              f"{N_SCAN}-row scan, {N_MAP}-row map x 387 columns fp32, three chained voxelisations (one kernel launch each), "
                                              "descriptor search, 50000-iteration RANSAC; the scene's map kept between scans (warm)")
        # the shape of a raw sensor scan: 60 000 points down to ~2 x 10^3 searched rows
        pp6 = synth.make_pair(60000, N_MAP, DIM, seed=11)
        raw6 = np.c_[pp6["q_xyz"], pp6["q_desc"]].astype(np.float32)
        map6 = np.c_[pp6["b_xyz"], pp6["b_desc"]].astype(np.float32)
        api6, poses6 = time_api(map6, raw6)
        api6["pose_err_vs_planted"] = float(np.linalg.norm(poses6[1] - pp6["T_gt"]))
        out["API_ransac_registration"]["raw_scan_60000"] = dict(api6, workload=f"the same call with a 60000-row raw scan against a {N_MAP}-row map")
        del pp6, raw6, map6
        del voxel_map, raw_scan, pp
    except Exception as e:
        out["API_ransac_registration"] = {"error": f"{type(e).__name__}: {e}"}
    # ---- C5
    n5, m5, d5 = 50000, 1000000, 768
    p5 = synth.make_pair_device(n5, m5, d5, seed=1, device=dev)
    a, b = C.c_void_p(), C.c_void_p()
    lib.vfm_prof_events_create(C.byref(a), C.byref(b))
    ts = []
    ms = C.c_float()
    pipe5 = RegistrationPipeline(n5, m5, d5, n_iter=RANSAC_ITERS, device=dev)
    for k in range(4):  # the coarse kernel of the pipeline's own search (gated family, best-score records on this data)
        lib.vfm_prof_arm(a, b)
        pipe5.register(p5["q_desc"], p5["q_xyz"], p5["b_desc"], p5["b_xyz"])
        lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
        if k:
            ts.append(ms.value)
    lib.vfm_prof_events_destroy(a, b)
    t5 = sorted(ts)[len(ts) // 2]
    t5_reg = timed(lambda: pipe5.register(p5["q_desc"], p5["q_xyz"], p5["b_desc"], p5["b_xyz"]), reps=3)
    r5 = pipe5.register(p5["q_desc"], p5["q_xyz"], p5["b_desc"], p5["b_xyz"])
    torch.cuda.synchronize()
    half5 = bool(pipe5.use_i8 and pipe5.half)   # the half-width pass: the kernel runs over the first 384 of the 768 columns
    half5_fp6 = half5 and bool(getattr(pipe5, "mx6_half", False))
    f5 = 2.0 * n5 * m5 * (d5 // 2 if half5 else d5)
    peak5 = MFMA_F6_PEAK_TFLOPS if half5_fp6 else MFMA_I8_PEAK_TOPS
    out["C5"] = {"workload": "50000-pt scan vs 1000000-pt map, 768-D, 50000 RANSAC iterations (one registration, serial)",
                 "ms_registration": t5_reg, "ms_coarse_kernel": t5, "correspondences": int(r5["count"].item()),
                 "pose_err_vs_planted": float(np.linalg.norm(r5["T"].cpu().numpy() - p5["T_gt"])),
                 "coarse_pass": pass_name(pipe5),
                 "roofline": {"bound": "mfma", "kernel": ("match_coarse_mx6q2_kernel<6, MX6_FUSE, false, 12, 4> (fp6 e2m3 32x32x64 scaled MFMA over the first 384 of 768 columns, 64 resident "
                                                          "queries per wave)" if half5_fp6
                                                          else "match_coarse_i8q2_kernel<12> (int8 32x32x32 MFMA over the first 384 of 768 columns, 64 resident queries per wave)" if half5
                                                          else "match_coarse_i8_kernel<24, 2> (int8 32x32x32 MFMA)"), "flops": f5,
                              "achieved": f5 / (t5 * 1e-3) / 1e12, "peak": peak5, "unit": "TFLOP/s",
                              "frac": f5 / (t5 * 1e-3) / 1e12 / peak5}}
    if half5_fp6:   # the same registration with the half-width pass on the int8 image (round 2's mode), for comparison
        del pipe5
        pipe5 = RegistrationPipeline(n5, m5, d5, n_iter=RANSAC_ITERS, device=dev, coarse="int8-half")
        lib.vfm_prof_events_create(C.byref(a), C.byref(b))
        ts = []
        for k in range(3):
            lib.vfm_prof_arm(a, b)
            pipe5.register(p5["q_desc"], p5["q_xyz"], p5["b_desc"], p5["b_xyz"])
            lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
            if k:
                ts.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
        out["C5"]["int8_half_width"] = {"ms_coarse_kernel": sorted(ts)[len(ts) // 2],
                                        "ms_registration": timed(lambda: pipe5.register(p5["q_desc"], p5["q_xyz"], p5["b_desc"], p5["b_xyz"]), reps=3)}
    # BASELINE.json configs[4] "fp16 descriptor storage": the same registration with the map held in fp16 (include/vfmreg.h VFM_ROWS_F16: rows
    # widened to fp32 as the kernels load them; tests/test_gpu_f16rows.py, test_c5_solve_at_full_size: the result of the widened rows, bit
    # for bit); the pair with fp32 storage is the line above
    try:
        del pipe5
        b16 = p5["b_desc"].half().contiguous()
        bw = b16.float().contiguous()
        pipe5 = RegistrationPipeline(n5, m5, d5, n_iter=RANSAC_ITERS, device=dev)
        for _ in range(3):
            pipe5.register(p5["q_desc"], p5["q_xyz"], b16, p5["b_xyz"])
            pipe5.synchronize()
            torch.cuda.synchronize()
            pipe5._poll_feedback()
        t16 = timed(lambda: pipe5.register(p5["q_desc"], p5["q_xyz"], b16, p5["b_xyz"]), reps=3)
        r16 = pipe5.register(p5["q_desc"], p5["q_xyz"], b16, p5["b_xyz"])
        torch.cuda.synchronize()
        T16 = r16["T"].clone()
        tw = timed(lambda: pipe5.register(p5["q_desc"], p5["q_xyz"], bw, p5["b_xyz"]), reps=3)
        rw = pipe5.register(p5["q_desc"], p5["q_xyz"], bw, p5["b_xyz"])
        torch.cuda.synchronize()
        out["C5"]["fp16_descriptor_storage"] = {
            "ms_registration": t16, "ms_registration_same_rows_stored_as_fp32": tw, "map_bytes": int(b16.numel() * 2), "map_bytes_fp32": int(bw.numel() * 4),
            "coarse_pass": pass_name(pipe5), "correspondences": int(r16["count"].item()),
            "pose_err_vs_planted": float(np.linalg.norm(T16.cpu().numpy() - p5["T_gt"])),
            "pose_equals_the_widened_rows_registration": bool(torch.equal(T16, rw["T"]))}
        del b16, bw
    except Exception as e:
        out["C5"]["fp16_descriptor_storage"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def timed_loop(lib, pipe, pairs, steps, warmup, settle=0):
    """The timed region's own form on one rank, for the figures under `extra`: `settle` registrations one at a time (the
    "auto" policy reads its feedback between them), `warmup` pipelined ones, then `steps` registrations back to back between
    two synchronisations; HIP events around every coarse kernel.  Returns (registrations/s, ms per step, mean coarse-kernel ms)."""
    import torch
    main = torch.cuda.current_stream()
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record(main)

    def reg(i):
        pr = pairs[i % len(pairs)]
        return pipe.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"], want_mask=True,
                             inputs_ready=ready if pipe.overlap else None)
    for i in range(settle):
        reg(i)
        pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
    events = []
    for _ in range(steps):
        a, b = C.c_void_p(), C.c_void_p()
        lib.vfm_prof_events_create(C.byref(a), C.byref(b))
        events.append((a, b))
    import gc
    gc.collect()   # (before the warm-up: behind it the GPU sat idle for the collection's length and the timed steps paid the clock ramp)
    gc.disable()
    for i in range(max(warmup, 1)):
        reg(i)
    pipe.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        lib.vfm_prof_arm(events[i][0], events[i][1])
        out = reg(i)
    pipe.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    ms = C.c_float()
    durs = []
    for a, b in events:
        lib.vfm_prof_elapsed_ms(a, b, C.byref(ms))
        durs.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
    return steps / dt, 1e3 * dt / steps, sum(durs) / len(durs), out


def pass_name(pipe):
    if not pipe.use_i8:
        return "fp16"
    if getattr(pipe, "mx6", False):
        return ("fp6 (MX e2m3), full width, packed top-2 records (VFM_RECORDS_MX6_TOP2)" if getattr(pipe, "mx6_top2", False)
                else "fp6 (MX e2m3), full width, gate test in the epilogue, no records (VFM_RECORDS_MX6_FUSED)" if getattr(pipe, "mx6_fused", False)
                else "fp6 (MX e2m3), full width, best-score records (VFM_RECORDS_MX6)")
    if pipe.half and getattr(pipe, "mx6_half", False):
        return ("fp6 (MX e2m3), half-width, survivor-only epilogue (VFM_RECORDS_MX6_HALF_FUSED)" if pipe._records() == 8
                else "fp6 (MX e2m3), half-width (VFM_RECORDS_MX6_HALF)")
    return "int8, half-width (VFM_RECORDS_HALF)" if pipe.half else ("int8, packed top-2 records" if pipe.top2 else "int8, best-score records")


def roofline_of(pipe, n, m, d, coarse_ms):
    """MFMA roofline entry of the coarse kernel a pipeline ran: operations AS LAUNCHED over the mean launch duration."""
    kcols = d // 2 if (pipe.use_i8 and pipe.half) else d
    flops = 2.0 * n * m * kcols
    peak = MFMA_F6_PEAK_TFLOPS if (getattr(pipe, "mx6", False) or (pipe.half and getattr(pipe, "mx6_half", False))) else (MFMA_I8_PEAK_TOPS if pipe.use_i8 else MFMA_F16_PEAK_TFLOPS)
    return {"bound": "mfma", "flops_per_launch": flops, "avg_launch_ms": coarse_ms, "achieved": flops / (coarse_ms * 1e-3) / 1e12,
            "peak": peak, "unit": "TFLOP/s", "frac": flops / (coarse_ms * 1e-3) / 1e12 / peak,
            "columns_multiplied": kcols, "all_pairs_product_flops": 2.0 * n * m * d}


HBM_PEAK_TBS = 8.0       # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # vector fp64 (~79)


def stage_table(dev, lib, pair, iters, records_kind):
    """every stage of a C2 registration ALONE on the GPU (HIP events on the stream it is launched on, median of 9),
    with the algorithmic work of D.3 and the fraction of the peak that bounds it.  The stages are the C-ABI calls the pipeline makes,
    at the record kind the timed region ran; in the pipeline they overlap (the headline is not their sum)."""
    import torch
    from vfmreg import _lib, ops
    q, b, q_xyz, b_xyz = pair["q_desc"], pair["b_desc"], pair["q_xyz"], pair["b_xyz"]
    n, d = q.shape
    m = b.shape[0]
    u8 = torch.uint8
    qb = torch.empty(lib.vfm_match_prepared_bytes(n, d), dtype=u8, device=dev)
    bb = torch.empty(lib.vfm_match_prepared_bytes(m, d), dtype=u8, device=dev)
    ws = torch.empty(lib.vfm_match_search_workspace_bytes(n, m, d), dtype=u8, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    sim = torch.empty(n, dtype=torch.float32, device=dev)
    gate = 0.8
    flags = (8 | 16) if records_kind in (7, 8) else 8 if records_kind in (5, 6, 9, 10) else 0   # VFM_PREPARE_MX6 (| _MX6_HALF)

    def st():
        return torch.cuda.current_stream().cuda_stream

    def prep():
        _lib.check(lib.vfm_match_prepare2_gated_p(b.data_ptr(), m, bb.data_ptr(), q.data_ptr(), n, qb.data_ptr(), d, flags, st()))

    def coarse():
        _lib.check(lib.vfm_match_search_coarse_gated_g(qb.data_ptr(), n, bb.data_ptr(), m, d, ws.data_ptr(), ws.numel(), records_kind, gate, st()))

    def finish():
        _lib.check(lib.vfm_match_search_finish_gated_r(q.data_ptr(), qb.data_ptr(), n, b.data_ptr(), bb.data_ptr(), m, d, idx.data_ptr(),
                                                       sim.data_ptr(), ws.data_ptr(), ws.numel(), gate, records_kind, st()))
    tc = {}

    def compact():
        tc["r"] = ops.threshold_compact(sim, idx, gate, q_xyz, b_xyz)
    ro = {}

    rws = {}

    def ransac():
        r = tc["r"]
        if "ws" not in rws:
            rws["ws"] = torch.empty(lib.vfm_ransac_workspace_bytes(r["corres"].shape[0], iters), dtype=torch.uint8, device=dev)
        ro["o"] = ops.ransac_corr(q_xyz, b_xyz, r["corres"], 10000.0, iters, seed=42, count=r["count"], out=ro.get("o"), ws=rws["ws"])

    def med(fn, before=(), reps=9):
        ts = []
        for i in range(reps + 2):
            for f in before:      # (coarse and finish consume the workspace: re-run what feeds them, untimed)
                f()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            e.record()
            e.synchronize()
            if i >= 2:
                ts.append(a.elapsed_time(e))
        return sorted(ts)[len(ts) // 2]
    t_prep = med(prep)
    t_coarse = med(coarse)
    t_finish = med(finish, before=(coarse,))
    t_compact = med(compact)
    t_ransac = med(ransac)
    C = int(tc["r"]["count"].item())
    half = records_kind in (3, 7, 8)
    kcols = d // 2 if half else d
    peak_mm = MFMA_F6_PEAK_TFLOPS if records_kind in (5, 6, 7, 8, 9) else MFMA_I8_PEAK_TOPS
    moved_prep = 4.0 * (n + m) * d + (n + m) * d * (1.0 + (0.375 if flags & 16 else 0.75 if flags else 0.0)) + n * d   # fp32 rows in; int8 image, fp6 image (half), scan's row-major int8 copy out
    rows = {
        "prepare (normalise rows, int8 + fp6 images)": {
            "ms": t_prep, "bound": "hbm", "algorithmic_bytes": 8.0 * (n + m) * d, "bytes_this_kernel_moves_once": moved_prep,
            "achieved_TBs": moved_prep / (t_prep * 1e-3) / 1e12, "peak_TBs": HBM_PEAK_TBS, "frac": moved_prep / (t_prep * 1e-3) / 1e12 / HBM_PEAK_TBS,
            "note": "SURVEY 8 D.3's figure (fp32 in, fp32 out) is 8 (N + M) D; the kernel writes one-byte and 6-bit images instead; since round 6 "
                    "(prep_once_kernel, DESIGN.md R6) every row is read ONCE -- the fraction counts what must move once"},
        "coarse pass": {
            "ms": t_coarse, "bound": "mfma", "flops": 2.0 * n * m * kcols, "achieved_TFLOPs": 2.0 * n * m * kcols / (t_coarse * 1e-3) / 1e12,
            "peak_TFLOPs": peak_mm, "frac": 2.0 * n * m * kcols / (t_coarse * 1e-3) / 1e12 / peak_mm},
        "finish (bin survivors, int8 rescan on the matrix cores, fp32 refinement, fp64 decision)": {
            "ms": t_finish, "bound": "latency (dependent round trips per surviving query; DESIGN.md R4.5)", "frac": None},
        "threshold + compact (VHM.cpp:501-511, 587-600)": {
            "ms": t_compact, "bound": "hbm", "algorithmic_bytes": 12.0 * n + 72.0 * C,
            "achieved_TBs": (12.0 * n + 72.0 * C) / (t_compact * 1e-3) / 1e12, "peak_TBs": HBM_PEAK_TBS,
            "frac": (12.0 * n + 72.0 * C) / (t_compact * 1e-3) / 1e12 / HBM_PEAK_TBS, "note": "a launch-latency object (1 MB); timed through ops.threshold_compact, which allocates its outputs on the host side -- the "
                                                      "kernel itself is 11-18 us in profiles/r04_bench_kernel_stats.csv"},
        "RANSAC + Kabsch (50 000 hypotheses, fp64)": {
            "ms": t_ransac, "bound": "latency (dependent launches; fp64 valu for what is executed)", "correspondences": C,
            "reference_flops": iters * (27.0 * C + 400.0), "peak_TFLOPs": FP64_VALU_PEAK_TFLOPS, "frac": None,
            "note": "reference_flops = SURVEY 8 D.3's count of the reference's work (every hypothesis scored over every correspondence); the kernels "
                    "bound every hypothesis first and run the oracle's fp64 arithmetic for the survivors only (DESIGN.md 4.3): no roofline fraction is "
                    "quoted against work that was avoided (VERDICT r5) -- `executed` counts what ran"},
    }
    try:   # the same stage on the operations it EXECUTED (the closed-form moment bound per hypothesis, the point-wise
        # fp32 pass where it was needed, the oracle-order fp64 scoring of the candidates)
        import ctypes as _ct   # (`C` is the correspondence count in this function)
        Cn = float(C)
        cnt = (_ct.c_int32 * 3)()
        _lib.check(lib.vfm_debug_ransac_counts(rws["ws"].data_ptr(), tc["r"]["corres"].shape[0], iters, cnt))
        n64 = iters if cnt[1] else int(cnt[0])
        f64 = n64 * (27.0 * Cn + 400.0) + iters * 400.0          # candidates over every correspondence + a Kabsch and a moment bound per hypothesis
        f32 = (iters * 27.0 * Cn) if cnt[2] else 0.0
        rows["RANSAC + Kabsch (50 000 hypotheses, fp64)"]["executed"] = {
            "hypotheses_scored_in_fp64": n64, "candidate_list_overflowed": bool(cnt[1]), "pointwise_fp32_pass": bool(cnt[2]),
            "fp64_flops": f64, "fp32_flops": f32, "frac_fp64_valu": f64 / (t_ransac * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
            "note": "a latency chain of 7 short launches at this size, not an ALU-bound kernel: the fraction says so"}
    except Exception as e:
        rows["RANSAC + Kabsch (50 000 hypotheses, fp64)"]["executed"] = {"error": f"{type(e).__name__}: {e}"}
    rows["sum_of_stages_ms"] = t_prep + t_coarse + t_finish + t_compact + t_ransac
    return rows


def c2_variants(dev, lib, pairs, steps, warmup, iters, streams):
    """Driver-timed C2 figures that do not rest on D.2's prunable noise, same pipeline construction and the
    same timed-loop form as the headline, OUTSIDE the headline's timed region:
      C2_full_width  coarse="int8": the all-pairs int8 product over all 384 columns (data independent);
      C2_sustained   the headline's configuration over 200 steps (the 20-step region carries ~2 ms of fill / drain / clock transient);
      C2_lifted      descriptors that look like lifted ViT features (every map row a bilinear sample of 16 x 21 patch grids of 60
                     images sharing a scene, + a common component): the mode the auto policy chooses, survivors per query."""
    import numpy as np
    import torch
    from vfmreg import synth
    from vfmreg.pipeline import RegistrationPipeline
    n, d = pairs[0]["q_desc"].shape
    m = pairs[0]["b_desc"].shape[0]
    out = {}

    def build(coarse):
        return RegistrationPipeline(n, m, d, n_iter=iters, device=dev, overlap_ransac=(streams == 2), overlap_prepare=(streams == 2),
                                    solve_streams=2, coarse=coarse)
    pipe = build("int8")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, steps, warmup)
    out["C2_full_width"] = {"workload": "C2, D.2 pairs, coarse pass pinned to the full-width int8 kernel (best-score records): 2 N M D per launch",
                            "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                            "roofline": roofline_of(pipe, n, m, d, cms)}
    del pipe
    pipe = build("mx6")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, steps, warmup)
    out["C2_full_width_mx6"] = {"workload": "C2, D.2 pairs, coarse pass pinned to the full-width fp6 kernel (MX e2m3 on the scaled MFMA, best-score "
                                            "records; operands prepared with the fp6 image as well): 2 N M D per launch",
                                "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                                "roofline": roofline_of(pipe, n, m, d, cms)}
    del pipe
    pipe = build("mx6-fused")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, steps, warmup)
    out["C2_full_width_mx6_fused"] = {"workload": "C2, D.2 pairs, the full-width fp6 kernel with the gate test in its epilogue (VFM_RECORDS_MX6_FUSED, round 5): 2 N M D "
                                                  "per launch, no record array and no selection sweep -- every column is multiplied, but what the kernel lists "
                                                  "depends on how many rows reach the gate (D.2: the planted match; maps with many rows above the gate overflow "
                                                  "the lists and take the guard's full-width int8 pass)",
                                      "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                                      "roofline": roofline_of(pipe, n, m, d, cms)}
    del pipe
    pipe = build("mx6-half")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, steps, warmup)
    out["C2_half_width_mx6"] = {"workload": "C2, D.2 pairs, coarse pass pinned to the half-width pass in fp6 (VFM_RECORDS_MX6_HALF: the headline's bound on "
                                            "the scaled MFMA, N M D operations per launch; the preparation writes the fp6 image too)",
                                "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                                "roofline": roofline_of(pipe, n, m, d, cms),
                                "note": "what `auto` settles on for D.2 data since the end of round 3 (DESIGN.md 0.11)"}
    del pipe
    pipe = build("int8-half")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, steps, warmup)
    out["C2_half_width_int8"] = {"workload": "C2, D.2 pairs, coarse pass pinned to the half-width pass on the int8 image (VFM_RECORDS_HALF: round 2's headline mode)",
                                 "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                                 "roofline": roofline_of(pipe, n, m, d, cms)}
    del pipe
    pipe = build("auto")
    v, msps, cms, _ = timed_loop(lib, pipe, pairs, 200, warmup, settle=4)
    out["C2_sustained"] = {"workload": "C2, D.2 pairs, the headline's pipeline over 200 timed steps", "value": v, "unit": "registrations/s",
                           "steps": 200, "ms_per_step": msps, "coarse_pass": pass_name(pipe), "roofline": roofline_of(pipe, n, m, d, cms),
                           "half_width_survivors_per_query": (pipe.last_rescans / n) if (pipe.half and pipe.last_rescans is not None) else None}
    del pipe
    torch.cuda.empty_cache()
    lifted = [synth.make_lifted_pair_device(n, m, d, seed=42 + p, device=dev, clouds=10, view_noise=0.1, common=1.0) for p in range(2)]
    pipe = build("auto")
    v, msps, cms, res = timed_loop(lib, pipe, lifted, steps, warmup, settle=6)
    errs = float(np.linalg.norm(res["T"].cpu().numpy() - lifted[(steps - 1) % 2]["T_gt"]))
    out["C2_lifted"] = {"workload": "C2 sizes, descriptors that look like lifted ViT features: every map row a bilinear sample of the 16 x 21 patch "
                                    "grids of 60 images (10 clouds x 6 cameras sharing a scene, view noise 0.1) plus a common component of 1 rms "
                                    "(background cosines ~0.5); scan = matched rows + 0.3 rms noise, 50 % outlier rows",
                        "value": v, "unit": "registrations/s", "steps": steps, "ms_per_step": msps, "coarse_pass": pass_name(pipe),
                        "roofline": roofline_of(pipe, n, m, d, cms), "correspondences": int(res["count"].item()), "pose_err_vs_planted": errs,
                        "half_width_probe_survivors_per_query": (pipe.last_probe / n) if pipe.last_probe is not None else None,
                        "rescanned_chunks_per_query": (pipe.last_rescans / n) if pipe.last_rescans is not None else None}
    del pipe, lifted
    torch.cuda.empty_cache()
    # row A6 (north_star's "all-pairs L2 + mutual-NN", registration_node.py:482-538) at C2 size on pair 0's descriptors
    from vfmreg import ops

    def ms_of(fn, reps=5):
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts[1:])[len(ts[1:]) // 2], r
    a, b = pairs[0]["q_desc"], pairs[0]["b_desc"]
    t_pairs, r = ms_of(lambda: ops.match_mutual_pairs(a, b))
    t_full, _ = ms_of(lambda: ops.match_mutual_l2(a, b), reps=3)
    flops = 2.0 * n * m * d + 2.0 * n * n * d      # the forward all-pairs product + the reverse one restricted to the matched rows
    out["A6_mutual_l2"] = {"workload": "find_correspondences(mutual_filter=True) (registration_node.py:482-538) at C2 size: exact Euclidean 1-NN of "
                                       "every scan row among the map, reverse direction at the matched map rows, mutual pairs; int8 MFMA "
                                       "coarse pass on the norm-sorted map, fp64 decision on the original rows",
                           "ms_mutual_pairs": t_pairs, "mutual_pairs": int(r[2].item()), "registrations_per_s_equivalent": 1e3 / t_pairs,
                           "ms_both_directions_all_rows": t_full,
                           "roofline": {"bound": "mfma", "flops": flops, "achieved": flops / (t_pairs * 1e-3) / 1e12, "peak": MFMA_I8_PEAK_TOPS,
                                        "unit": "TFLOP/s", "frac": flops / (t_pairs * 1e-3) / 1e12 / MFMA_I8_PEAK_TOPS,
                                        "note": "whole call (norms, sort, operand preparation, both coarse passes, selection, rescans, fp64 decision) "
                                                "over the int8 peak"}}
    return out


RESIDENT_MAX = 32  # distinct scene pairs kept in HBM per rank (338 MB each); longer runs cycle through them


def run_c3_form(args, dev, rank, world):
    """`--form c3`: the job's scene pairs end to end from uint8 images, sharded pair p -> rank p mod N like the c2 form:
    every rank drives one EndToEndPipeline over its resident pairs, no data-path collective, one gather of the poses at the end."""
    import gc
    import numpy as np
    import torch
    import torch.distributed as dist
    from vfmreg import dist as vdist, ops
    from vfmreg import vit as V
    from vfmreg.pipeline import EndToEndPipeline
    n, m = args.n, args.m
    B, H, W = 6, 1200, 1600
    num_pairs = args.pairs if args.pairs > 0 else world * args.steps
    if 0 < args.pairs < world:
        raise SystemExit(f"--pairs {args.pairs} is smaller than the number of GPUs ({world})")
    mine = vdist.shard_pairs(num_pairs, rank, world)
    steps = len(mine) if args.pairs > 0 else args.steps
    n_res = max(1, min(len(mine), 8))   # resident pairs per GPU (6 images + a 200 000 x 384 map each: 0.35 GB)
    model = V.ViTS14(V.random_weights(0), H, W, device=dev)
    K = np.array([[800.0, 0, 800], [0, 800, 600], [0, 0, 1]])
    Ps = []
    for i in range(6):
        y = np.deg2rad(60 * i)
        R = np.stack([[np.sin(y), -np.cos(y), 0], [0, 0, -1], [np.cos(y), np.sin(y), 0]])
        Ps.append(K @ np.c_[R, np.zeros(3)])
    rig = [dict(mode=ops.PROJ_KITTI, mats=[Ps[c]], fc=None, subsample=1.0, win=None, H=H, W=W, rot_mode=0) for c in range(6)]
    pairs = []
    for j in range(n_res):   # generated ON the owning rank from seed 42 + global pair id, resident before the timed region
        rng = np.random.default_rng(42 + mine[j])
        imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).to(dev)
        xyz = np.c_[rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-2, 6, n)]
        pcl = torch.from_numpy(np.ascontiguousarray(np.insert(xyz, 3, 1, axis=1).T)).to(dev)
        grids = model.forward(imgs)
        desc = torch.empty((n, 384), dtype=torch.float32, device=dev)
        filled = torch.zeros(n, dtype=torch.uint8, device=dev)
        ops.LiftPlan([dict(c, proj_image=None, grid=grids[k], Hup=H, Wup=W, raw_image=imgs[k]) for k, c in enumerate(rig)], 384)(pcl, desc, filled)
        g = torch.Generator(device=dev).manual_seed(3 + mine[j])
        b_desc = torch.randn(m, 384, device=dev, generator=g)
        pick = torch.randperm(m, device=dev, generator=g)[:n]
        b_desc[pick] = desc + 0.02 * desc.abs().mean() * torch.randn(n, 384, device=dev, generator=g)
        b_xyz = torch.rand(m, 3, device=dev, generator=g, dtype=torch.float64) * 100.0
        q_xyz = torch.from_numpy(np.ascontiguousarray(xyz)).to(dev)
        b_xyz[pick] = q_xyz + 0.02 * torch.randn(n, 3, device=dev, generator=g, dtype=torch.float64)
        pairs.append(dict(imgs=imgs, pcl=pcl, q_xyz=q_xyz, b_desc=b_desc, b_xyz=b_xyz))
    G = max(int(args.feature_group), 1)
    e2e = EndToEndPipeline(model, rig, n, m, n_iter=args.iters, depth=4, device=dev, group=G, group_depth=3)
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream())
    res_T = torch.empty((steps, 4, 4), dtype=torch.float64, device=dev)
    res_c = torch.empty((steps, 1), dtype=torch.int64, device=dev)

    def group_step(lo, hi, keep):
        def snap(k, out):
            if keep:
                with torch.cuda.stream(out["result_stream"]):
                    res_T[lo + k].copy_(out["T"])
                    res_c[lo + k].copy_(out["count"])
        ps = [pairs[i % n_res] for i in range(lo, hi)]
        e2e.submit_group([(p["imgs"], p["pcl"], p["q_xyz"], p["b_desc"], p["b_xyz"]) for p in ps], inputs_ready=ready, on_result=snap)

    def step(i, keep=None):
        p = pairs[i % n_res]
        out = e2e.submit(p["imgs"], p["pcl"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ready)
        if keep is not None:
            with torch.cuda.stream(out["result_stream"]):
                res_T[keep].copy_(out["T"])
                res_c[keep].copy_(out["count"])
    vdist.gather_poses(torch.zeros((1, 4, 4), dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev), world, rank, world)
    gc.collect()
    gc.disable()
    for i in range(8 + max(args.warmup, 1)):   # the policy settles (one registration at a time), then the pipelined warm-up
        step(i)
        e2e.reg._poll_feedback()
        if i < 8:
            e2e.synchronize()
            torch.cuda.synchronize()
    if G > 1:
        for lo in range(0, 2 * G, G):           # ... and two groups through the batch path
            group_step(lo, lo + G, keep=False)
            e2e.reg._poll_feedback()
    e2e.synchronize()
    grouped = dist.is_available() and dist.is_initialized()
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if G > 1:
        for lo in range(0, steps, G):
            group_step(lo, min(lo + G, steps), keep=True)
    else:
        for i in range(steps):
            step(i, keep=i)
    e2e.synchronize()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    all_poses, all_counts = vdist.gather_poses(res_T[:steps], res_c[:steps].reshape(-1), num_pairs, rank, world)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_rank = [steps / local_elapsed]
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        vdist.all_reduce_max(tmax)
        elapsed = float(tmax.item())
        rates = torch.zeros(world, dtype=torch.float64, device=dev)
        vdist.all_gather_rows(rates, torch.tensor([steps / local_elapsed], dtype=torch.float64, device=dev))
        per_rank = [float(x) for x in rates.cpu()]
    local_ids = mine[:steps] if args.pairs > 0 else [rank + world * i for i in range(steps)]
    errs = [float(np.linalg.norm(all_poses[g].cpu().numpy() - np.eye(4))) for g in local_ids]   # the planted transform is the identity
    if rank == 0 and args.dump_poses:
        np.savez(args.dump_poses, poses=all_poses.cpu().numpy(), counts=all_counts.cpu().numpy())
    if rank == 0:
        print(json.dumps({
            "metric": "registrations/sec from uint8 images (C3: ViT-S/14 on 6 x 1200x1600 + lifting + 20k<->200k registration)",
            "value": num_pairs / elapsed, "unit": "registrations/s", "n_gpus": world, "steps": vdist.pairs_per_rank(num_pairs, world),
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / vdist.pairs_per_rank(num_pairs, world), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp16 ViT (fp32 accumulation) + the c2 form's matcher and f64 RANSAC", "data": "synthetic",
            "config": {"workload": f"C3 form: {num_pairs} scene pair(s) end to end from uint8 images, pair p -> rank p mod N, {n_res} resident per GPU; "
                                   f"{n}-pt scan vs {m}-pt map, {args.iters} RANSAC iterations", "scene_pairs_total": num_pairs,
                       "records_kind": int(e2e.reg._records()), "max_pose_err_vs_planted": max(errs), "pairs_per_vit_call": G,
                       "correspondences_last_step": int(all_counts[local_ids[-1]].item()),
                       "per_rank_registrations_per_s": per_rank,
                       "collective": f"one all_gather_into_tensor of the poses ({dist.get_backend()})" if grouped else "none (single process, no launcher)"},
            "per_rank_registrations_per_s": per_rank, "roofline": None, "cpu_baseline": None}), flush=True)




def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed registrations per GPU")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=0,
                    help="config C4: total number of independent scene pairs of the job, sharded pair p -> rank p mod N "
                         "(overrides --steps: every rank registers its ceil(pairs / N) pairs once)")
    ap.add_argument("--resident", type=int, default=RESIDENT_MAX,
                    help="distinct scene pairs kept in HBM per rank (338 MB each at C2 size); a rank with more pairs than that cycles through "
                         "its resident ones (pair j of the rank is registered on the data of its pair j mod resident)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C5 measurements reported under `extra`")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC passes under profiles/ instead of two rocprofv3 --pmc subprocesses of this run "
                         "(--no-extra implies it; so does running bench.py itself under rocprofv3)")
    # (--scan-rows / --map-rows: the spellings to use behind `python -m torch.distributed.run`, whose own parser takes "--n" / "--m" for
    # abbreviations of its options even behind the script's name)
    ap.add_argument("--n", "--scan-rows", dest="n", type=int, default=N_SCAN)
    ap.add_argument("--m", "--map-rows", dest="m", type=int, default=N_MAP)
    ap.add_argument("--iters", type=int, default=RANSAC_ITERS)
    ap.add_argument("--form", choices=("c2", "c3"), default="c2",
                    help="c2 (default, BASELINE.json's metric): descriptors resident in HBM; c3: every pair end to end from uint8 images "
                         "(ViT-S/14 on 6 x 1200x1600 + projection / lifting + registration, vfmreg.pipeline.EndToEndPipeline) -- with --pairs an "
                         "N-GPU job shards end-to-end pairs the same way (information only: the headline stays the c2 form)")
    ap.add_argument("--feature-group", type=int, default=1,
                    help="--form c3 only: the cameras of this many consecutive pairs of a rank go through the ViT in one call "
                         "(EndToEndPipeline.submit_group); 1 = one ViT call per pair")
    ap.add_argument("--streams", type=int, default=2,
                    help="2: RANSAC of pair i overlaps the matching of pair i+1 on a second HIP stream; 1: serial")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend under a launcher: nccl (= RCCL over xGMI, the default and what an N-GPU job runs) or gloo "
                         "(collectives staged through host memory: lets several ranks share ONE device, which RCCL refuses -- the form "
                         "tests/test_gpu_multirank.py drives the real path in on a one-GPU box)")
    ap.add_argument("--device-index", type=int, default=None,
                    help="HIP device of this rank (default: LOCAL_RANK); with --backend gloo every rank may name device 0")
    ap.add_argument("--dump-poses", default=None,
                    help="rank 0 writes the gathered poses / correspondence counts of the job (global pair order) to this .npz")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if args.device_index is None else args.device_index
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vfmreg import _lib, synth
    from vfmreg import dist as vdist
    from vfmreg.pipeline import RegistrationPipeline
    if os.environ.get("VFM_LAZY_STREAMS") == "1":   # A/B only (tools/ab_queue_touch.sh): side streams bound to hardware queues at first use
        from vfmreg import pipeline as _pl
        _pl.TOUCH_STREAMS_AT_CREATION = False

    rank, world = vdist.init_from_env(backend=args.backend, device=dev)  # "nccl" == RCCL on ROCm

    lib = _lib.load()
    if os.environ.get("VFM_VARIANT"):  # A/B runs (tools/r02_prof.sh): 4 = dense per-chunk records + select kernel
        lib.vfm_debug_set_coarse_variant(int(os.environ["VFM_VARIANT"]))
    if os.environ.get("VFM_PREP_GRID"):  # A/B runs: workgroups of the operand-preparation kernel (-1 = one per 128-row group)
        lib.vfm_debug_set_prep_grid(int(os.environ["VFM_PREP_GRID"]))
    if os.environ.get("VFM_SLICES"):   # A/B runs: number of map slices of the coarse pass (0 = heuristic)
        lib.vfm_debug_set_coarse_slices(int(os.environ["VFM_SLICES"]))
    n, m, d = args.n, args.m, DIM
    if args.form == "c3":
        run_c3_form(args, dev, rank, world)
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
        return
    # Global scene pairs: pair p runs on rank p mod world and is generated ON ITS OWNER from
    # seed 42 + p (D.2).  They are resident in HBM before the timed region starts.
    if 0 < args.pairs < world:
        raise SystemExit(f"--pairs {args.pairs} is smaller than the number of GPUs ({world})")
    num_pairs = args.pairs if args.pairs > 0 else world * args.steps
    mine = vdist.shard_pairs(num_pairs, rank, world)
    steps = len(mine) if args.pairs > 0 else args.steps
    n_res = max(1, min(len(mine), max(1, args.resident)))
    pairs = [synth.make_pair_device(n, m, d, seed=42 + mine[j], device=dev) for j in range(n_res)]
    # --streams 2 (default): pipeline over independent scene pairs (BASELINE config C4: "one per stream"):
    # operand preparation + the MFMA coarse pass of pair i+1 run on the main stream while the filter / exact decision /
    # RANSAC of pair i run on a side stream (vfmreg/pipeline.py).  Coarse passes never overlap each other, so the
    # HIP-event duration of the coarse kernel stays a per-launch figure.
    S = 2 if args.streams >= 2 else 1
    pipe = RegistrationPipeline(n, m, d, n_iter=args.iters, device=dev, overlap_ransac=(S == 2),
                                overlap_prepare=(S == 2 and os.environ.get("VFM_OVERLAP_PREPARE", "1") == "1"),
                                solve_streams=int(os.environ.get("VFM_SOLVE_STREAMS", "2")),
                                coarse=os.environ.get("VFM_COARSE", "auto"))  # A/B: "int8" / "int8-top2" / "fp16" fix the pass

    # (a high-priority matching stream was tried: no measurable difference)
    match_stream = torch.cuda.current_stream()

    torch.cuda.synchronize()
    inputs_ready = torch.cuda.Event()  # the resident scene pairs are complete from here on
    inputs_ready.record(match_stream)

    def step(i):
        p = pairs[i % n_res]
        with torch.cuda.stream(match_stream):
            return pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], want_mask=True,
                                 inputs_ready=inputs_ready if S == 2 else None)

    # everything the timed region needs is created BEFORE the warm-up, so that nothing but the barrier + synchronise the
    # contract asks for lies between the last warm-up step and the first timed one (an idle gap of a few milliseconds
    # drops the shader clock, and the first timed registrations then pay the ramp)
    events = []
    for _ in range(steps):
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
        events.append((a, b))

    res_T = torch.empty((steps, 4, 4), dtype=torch.float64, device=dev)
    res_c = torch.empty((steps, 1), dtype=torch.int64, device=dev)

    def register_pair(i):
        lib.vfm_prof_arm(events[i][0], events[i][1])
        out = step(i)
        with torch.cuda.stream(out["result_stream"]):  # snapshot the result on the producing stream
            res_T[i].copy_(out["T"])
            res_c[i].copy_(out["count"])

    # untimed warm-up: the sharding / gather path first, then the complete step
    vdist.gather_poses(torch.zeros((1, 4, 4), dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                       world, rank, world)
    import gc
    gc.collect()      # (the interpreter's cyclic collector stays out of the timed region: a pause there is tens of ms of a 15 ms run;
    gc.disable()      # collected BEFORE the warm-up: between warm-up and timed region it left the GPU idle for its whole length)
    # The `auto` policy picks the coarse pass from measurements of the data it is fed (a probe of the half-width pass on the first
    # registration, then every search's own feedback): it is given SETTLE registrations, one at a time, to do so -- set-up, like
    # the extras' timed_loop(settle=6); with W <= 2 pipelined warm-up steps the switch used to fall into the timed region (1020
    # instead of 1330 registrations/s at --warmup 1).  Reported as config.policy_settle_registrations.
    settle = SETTLE if pipe.coarse == "auto" else 0
    settle_ms = []   # what each of them took, host clock, synchronised: the probe of the first one and the policy's switches are in here
    for i in range(settle):
        ts = time.perf_counter()
        step(i)
        with torch.cuda.stream(match_stream):
            pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
        settle_ms.append(round((time.perf_counter() - ts) * 1e3, 3))
    for i in range(int(os.environ.get("VFM_BENCH_PRECOND", "0"))):   # A/B only (tools/): extra untimed registrations in front of the warm-up
        step(i)
    for i in range(max(args.warmup, 1)):
        step(i)
    with torch.cuda.stream(match_stream):
        pipe.synchronize()
    torch.cuda.current_stream().wait_stream(match_stream)

    grouped = dist.is_available() and dist.is_initialized()  # launched through torch.distributed.run
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # every rank registers its pairs (no data-path collective), then ONE all_gather of the poses
    host_t = []
    for i in range(steps):
        register_pair(i)
        host_t.append(time.perf_counter() - t0)
    with torch.cuda.stream(match_stream):
        pipe.synchronize()
    torch.cuda.current_stream().wait_stream(match_stream)
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0   # this rank's own work, before the gather (stragglers show up here)
    # ragged --pairs (not a multiple of N): gather_poses pads the shorter ranks to ceil(pairs / N) rows
    all_poses, all_counts = vdist.gather_poses(res_T[:steps], res_c[:steps].reshape(-1), num_pairs, rank, world)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    per_rank = [steps / local_elapsed]
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        vdist.all_reduce_max(tmax)
        elapsed = float(tmax.item())
        rates = torch.zeros(world, dtype=torch.float64, device=dev)
        vdist.all_gather_rows(rates, torch.tensor([steps / local_elapsed], dtype=torch.float64, device=dev))
        per_rank = [float(x) for x in rates.cpu()]
    print(f"[rank {rank}] {steps} registrations in {local_elapsed * 1e3:.1f} ms = {steps / local_elapsed:.1f} registrations/s "
          f"(before the gather)", file=sys.stderr, flush=True)

    ms = C.c_float()
    durs = []
    for a, b in events:
        _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
        durs.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
    coarse_ms = sum(durs) / max(len(durs), 1)
    if os.environ.get("VFM_BENCH_TRACE"):  # per-step view of the timed region (start-up transient, host launch pace)
        print("[trace] coarse kernel ms per step: " + " ".join(f"{x:.3f}" for x in durs[:64]), file=sys.stderr)
        print("[trace] host time after each step's launches, ms: " + " ".join(f"{x * 1e3:.2f}" for x in host_t[:64]),
              file=sys.stderr, flush=True)

    hbm_peak = torch.cuda.max_memory_allocated(dev)   # resident pairs + buffer sets of this rank (config C4: 32 pairs -> ~11 GB of 288)
    mode_i8, mode_half, mode_top2 = bool(pipe.use_i8), bool(pipe.use_i8 and pipe.half), bool(pipe.use_i8 and pipe.top2 and not pipe.half)
    records_kind = int(pipe._records()) if pipe.use_i8 else 2   # include/vfmreg.h VFM_RECORDS_*: the kind the timed region ended in
    pipe._poll_feedback()
    surv = pipe.last_rescans
    # the same kernel without the concurrent RANSAC stream (information only; not part of `value`)
    iso = []
    if S == 2:
        pipe1 = RegistrationPipeline(n, m, d, n_iter=args.iters, device=dev, overlap_ransac=False,
                                     coarse=(("mx6-half" if getattr(pipe, "mx6_half", False) else "int8-half") if mode_half
                                             else "int8-top2" if mode_top2 else "int8" if mode_i8 else "fp16"))
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
        for i in range(6):
            pr = pairs[i % n_res]
            lib.vfm_prof_arm(a, b)
            pipe1.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
            _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
            if i:
                iso.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
        del pipe1
    iso_ms = (sum(iso) / len(iso)) if iso else coarse_ms

    # sanity of the timed work: every pose of this rank must recover the planted transform of ITS pair
    local_ids = mine[:steps] if args.pairs > 0 else [rank + world * i for i in range(steps)]
    errs = [float(np.linalg.norm(all_poses[g].cpu().numpy() - pairs[i % n_res]["T_gt"])) for i, g in enumerate(local_ids)]
    ncorr = int(all_counts[local_ids[-1]].item())
    T0 = all_poses[0].cpu().numpy()  # global pair 0 lives on rank 0
    if rank == 0 and args.dump_poses:
        np.savez(args.dump_poses, poses=all_poses.cpu().numpy(), counts=all_counts.cpu().numpy())

    if rank == 0:
        # which coarse pass ran: the int8 one for d = 256 ... 768 unless an A/B variant forces the fp16 pass; of the int8 pass the
        # kind the pipeline ended the timed region in (vfmreg/pipeline.py: the half-width pass over the first d / 2 columns while
        # few chunks survive it -- D.2 descriptors -- else best-score / packed top-2 records over all d)
        i8 = d in (256, 384, 512, 640, 768) and os.environ.get("VFM_VARIANT", "0") in ("0", "10", "12") and mode_i8
        half = i8 and mode_half
        half6 = half and bool(getattr(pipe, "mx6_half", False))   # the half-width pass on the fp6 image (VFM_RECORDS_MX6_HALF)
        kcols = d // 2 if half else d
        flops = 2.0 * n * m * kcols   # what the dominant kernel computes per launch
        achieved = flops / (coarse_ms * 1e-3) / 1e12
        peak = MFMA_F6_PEAK_TFLOPS if half6 else (MFMA_I8_PEAK_TOPS if i8 else MFMA_F16_PEAK_TFLOPS)
        fused6 = half6 and records_kind == 8
        ns3 = fused6 and d == 384   # (round 5: three 32-query tiles per wave in the fused half-width kernel at d = 384)
        kernel = ((f"match_coarse_mx6q2_kernel<{kcols // 64}, {'MX6_FUSE' if fused6 else 'MX6_BEST'}, false, {d // 64}, 4, 4, {3 if ns3 else 2}> (v_mfma_scale_f32_32x32x64_f8f6f4 on "
                   f"microscaled fp6 -- e2m3 elements, one power-of-two scale per 32 columns -- over the first {kcols} of {d} columns: the half-width pass "
                   "in fp6; the other half is bounded by Cauchy-Schwarz against the cosine gate, the image's quantisation by its measured residual "
                   "norms, and only surviving chunks are scored over all columns (int8 MFMA rescan, fp32 refinement, fp64 decision); " + ("96" if ns3 else "64") + " resident "
                   "queries per wave, " + ("survivors of the bound listed by the kernel itself (no records)" if fused6 else "one best-score record per (query, chunk)") + ")") if half6
                  else (f"match_coarse_i8q2_kernel<{kcols // 32}> (int8 32x32x32 MFMA over the first {kcols} of {d} columns -- the half-width pass: the "
                   "other half is bounded by Cauchy-Schwarz against the cosine gate and only surviving chunks are scored over all columns -- "
                   "64 resident queries per wave, exact integer scores, one best-score record per (query, chunk))") if half
                  else ("match_coarse_i8q2_kernel<12> (int8 32x32x32 MFMA, 64 resident queries per wave, exact integer scores, "
                        + ("packed top-2 records" if mode_top2 else "one best-score record per (query, chunk)") + ")") if i8
                  else "match_coarse_pipe_kernel<24, true> (fp16 32x32x16 MFMA, sparse row-level records)")
        traffic, traffic_src = None, None  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
        names = (("r04_pmc_match_coarse_mx6half.json", "r03_pmc_match_coarse_mx6half.json") if half6
                 else ("r03_pmc_match_coarse_i8half.json", "r02_pmc_match_coarse_i8half.json") if half
                 else ("r03_pmc_match_coarse_i8.json", "r02_pmc_match_coarse_i8.json") if i8
                 else ("r02_pmc_match_coarse_f16.json", "r01_pmc_match_coarse.json"))
        for name in names:
            pmc = ROOT / "profiles" / name
            if pmc.exists() and (n, m, d) == (N_SCAN, N_MAP, DIM):
                traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{name} (FETCH_SIZE / WRITE_SIZE from separate --pmc passes, corrected per MI355X_MICROARCH.md)"
                break
        traffic_committed = traffic
        profiled = any(k.startswith(("ROCP_", "ROCPROF", "ROCTRACER_")) for k in os.environ)   # bench.py itself under rocprofv3: no nested profiler
        if (world == 1 and not args.no_live_traffic and not args.no_extra and not profiled and (n, m, d) == (N_SCAN, N_MAP, DIM) and mode_i8):
            # ... and measured in THIS run: the same two passes on the same kernel, in subprocesses (this process holds no counters)
            live = live_traffic(records_kind)
            if live is not None:
                traffic = live["hbm_bytes_per_launch"]
                traffic_src = (f"this run: two separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE` passes of tools/prof_match.py at record "
                               f"kind {records_kind} ({live['launches']} launches of {live['kernel']}; FETCH_SIZE {live['FETCH_SIZE_KB']:.0f} KB doubled per "
                               f"MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE {live['WRITE_SIZE_KB']:.0f} KB; {live['seconds']:.0f} s); committed "
                               f"passes of the round: {traffic_committed} bytes")
        line = {
            "metric": "registrations/sec (20k<->200k pts, 384-D)", "value": num_pairs / elapsed,
            "unit": "registrations/s", "n_gpus": world, "steps": vdist.pairs_per_rank(num_pairs, world), "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / vdist.pairs_per_rank(num_pairs, world), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("fp6 (MX e2m3) coarse pass (scaled MFMA, fp32 scores + bounds from measured residuals) + int8 rescan (exact integer scores)" if half6
                      else "int8 coarse pass (MFMA, exact integer scores + proven quantisation bounds)" if i8 else "f16 coarse pass (MFMA)")
                     + " + f32 refinement + f64 exact decision / f64 RANSAC",
            "data": "synthetic",
            "config": {"workload": f"C2: {n}-pt scan vs {m}-pt map, {d}-D descriptors precomputed and resident in "
                                   f"HBM, {args.iters} RANSAC iterations, cosine >= 0.8; map renormalised every step",
                       "registrations_per_gpu": vdist.pairs_per_rank(num_pairs, world), "scene_pairs_total": num_pairs,
                       "resident_scene_pairs_per_gpu": n_res, "pair_seed": "42 + global pair id, generated on the owning rank",
                       "hbm_peak_allocated_gb": hbm_peak / 1e9,
                       "parallelism": f"{world} GPU shard(s) x {S} stream(s), independent scene pairs (pair p -> rank p mod N)",
                       "collective": (f"one all_gather_into_tensor of the poses ({dist.get_backend()}{' = RCCL' if dist.get_backend() == 'nccl' else ', staged through host memory'})" if grouped
                                      else "none (single process, no launcher)"),
                       "correspondences_last_step": ncorr, "max_pose_err_vs_planted": max(errs),
                       # what the parity claims rest on for the rows whose arithmetic lives in third-party code that is not
                       # under /root/reference
                       "parity_note": "indices / masks / pose are bit-equal to the repo's CPU oracle; for A1 (DINOv2 via FeatUp), A5's search (faiss "
                                      "IndexFlatIP + fvec_renorm_L2) and A8 (Open3D 0.18 RANSAC + Eigen::umeyama) that oracle RESTATES the published "
                                      "algorithms -- those dependencies are absent here, the reference holds no golden vectors for them, and RANSAC's "
                                      "thread-order-dependent generator is replaced by a counter-based one; A2 / A3 / A4 / A9 / print_errors / HDF5 are "
                                      "pinned to outputs of the imported reference (tests/golden)",
                       # untimed, in front of the W warm-up steps: registrations the auto policy reads its feedback between (set-up)
                       "policy_settle_registrations": settle,
                       # ... and how long each took, one at a time, synchronised (ms): the first carries the half-width probe, the lazy
                       # set-up of the kernels' attributes and the first launches; none of it is in the timed region
                       "policy_settle_ms": settle_ms,
                       # the record kind (include/vfmreg.h VFM_RECORDS_*) of the timed registrations: tests/test_gpu_bench_config.py
                       # compares exactly this kind with the oracle at this size (BENCH_RECORDS_KIND), tests/test_gpu_bench.py ties the two
                       "records_kind": records_kind,
                       "coarse_pass": (("fp6 (MX e2m3), half-width, survivor-only epilogue (VFM_RECORDS_MX6_HALF_FUSED)" if records_kind == 8
                                        else "fp6 (MX e2m3), half-width (VFM_RECORDS_MX6_HALF)") if half6 else "int8, half-width (VFM_RECORDS_HALF)" if half
                                       else "int8, packed top-2 records" if (i8 and mode_top2)
                                       else "int8, best-score records" if i8 else "fp16"),
                       # (query, chunk) pairs that survive the half-width bound, per query, in the last search the policy has read back:
                       # the figure the pruning rests on (D.2: the planted matches and nothing else, ~0.5; descriptors that are alike: hundreds)
                       "half_width_survivors_per_query": (surv / n) if (half and surv is not None) else None},
            "per_rank_registrations_per_s": per_rank,
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "peak_note": ("dense fp6 scaled MFMA, 10 PFLOP/s (the guide's spec at 2.4 GHz; bare MFMAs sustain 6.4-6.5 on this chip: "
                                       "tools/probe/mx6_probe.hip)" if half6
                                       else "dense int8 MFMA, integer multiply-adds counted as 2 operations each" if i8 else "dense fp16 MFMA"),
                         "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                         "flops_per_launch": flops, "avg_launch_ms": coarse_ms,
                         "flops_note": ("operations this kernel performs: 2 N M D/2 -- the reference's all-pairs product is 2 N M D = "
                                        f"{2.0 * n * m * d:.4g}; the half-width pass computes half of it and bounds the rest (DESIGN.md 4.15); "
                                        "achieved / frac count the operations performed, not the product avoided") if half else
                                       "2 N M D, the reference's all-pairs product",
                         "single_stream": {"avg_launch_ms": iso_ms, "achieved": flops / (iso_ms * 1e-3) / 1e12,
                                           "frac": flops / (iso_ms * 1e-3) / 1e12 / peak,
                                           "note": "same kernel without the RANSAC of the previous pair running beside it"}},
        }
        extra = {"note": "measured outside the timed region; the headline `value` is C2 only"}
        if not args.no_cpu_baseline and world == 1:
            host = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in pairs[0].items()}
            line["cpu_baseline"], delta = cpu_baseline(host, args.iters, T_gpu=T0)
            if delta is not None:
                extra["pose_delta_vs_oracle"] = delta
        else:
            line["cpu_baseline"] = None
        if not args.no_extra and world == 1 and (n, m) == (N_SCAN, N_MAP):  # N > 1: the other ranks would idle at the barrier
            del pipe
            try:
                extra.update(c2_variants(dev, lib, pairs[:4], vdist.pairs_per_rank(num_pairs, world), args.warmup, args.iters, S))
            except Exception as e:
                extra["error_c2_variants"] = f"{type(e).__name__}: {e}"
            try:
                extra["stages"] = stage_table(dev, lib, pairs[0], args.iters, records_kind)
            except Exception as e:
                extra["stages"] = {"error": f"{type(e).__name__}: {e}"}
            pairs.clear()
            torch.cuda.empty_cache()
            try:
                extra.update(extra_configs(dev))
            except Exception as e:  # never lose the headline line to an auxiliary measurement
                extra["error"] = f"{type(e).__name__}: {e}"
            try:   # the two stages of C3 in front of the registration, in the same table
                c3 = extra.get("C3", {})
                if "ms_vit" in c3 and isinstance(extra.get("stages"), dict) and "error" not in extra["stages"]:
                    extra["stages"]["ViT-S/14 on 6 x 1200x1600 (C3)"] = {
                        "ms": c3["ms_vit"], "bound": "mfma", "flops": c3["vit_roofline"]["flops"], "achieved_TFLOPs": c3["vit_roofline"]["achieved"],
                        "peak_TFLOPs": MFMA_F16_PEAK_TFLOPS, "frac": c3["vit_roofline"]["frac"],
                        "note": "63 dependent launches; per scan 0.30-0.36 ms when the cameras of 4-8 pairs share a call (extra.ViT_batched, C3_pipelined.grouped)"}
                    lift_bytes = 6 * N_SCAN * 24.0 + 4.0 * N_SCAN * DIM + 6 * 16 * 21 * DIM * 4.0
                    extra["stages"]["projection + lifting, 6 cameras (C3)"] = {
                        "ms": c3["ms_project_lift"], "bound": "hbm", "algorithmic_bytes": lift_bytes,
                        "achieved_TBs": lift_bytes / (c3["ms_project_lift"] * 1e-3) / 1e12, "peak_TBs": HBM_PEAK_TBS,
                        "frac": lift_bytes / (c3["ms_project_lift"] * 1e-3) / 1e12 / HBM_PEAK_TBS}
            except Exception as e:
                extra["stages"]["error_c3_rows"] = f"{type(e).__name__}: {e}"
        # the figure on descriptors that are alike (what real lifted ViT features look like) beside `value`
        if isinstance(extra.get("C2_lifted"), dict) and "value" in extra["C2_lifted"]:
            line["config"]["C2_lifted_registrations_per_s"] = extra["C2_lifted"]["value"]
            line["config"]["C2_lifted_note"] = ("same sizes and pipeline, descriptors that look like lifted ViT features (extra.C2_lifted): the half-width bound "
                                                "does not prune there and the full-width fp6 pass runs; `value` is SURVEY D.2's iid data")
        line["extra"] = extra
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
