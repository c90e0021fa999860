#!/usr/bin/env python
"""bench.py -- registrations/sec of the correspondence-and-solve hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1] ("C2"): one step = one registration of a 20 000-point scan
against a 200 000-point map with 384-D descriptors (precomputed, resident in HBM) and 50 000 RANSAC
iterations: normalise + fp16 fragment conversion of BOTH clouds (the reference renormalises the map
on every call, VoxelHashMap.cpp:469-482), exact top-1 inner-product search, cosine >= 0.8 threshold
and compaction, correspondence RANSAC with 3-point Kabsch.  Synthetic inputs of SURVEY.md 8 D.2.

Multi-GPU: independent scene pairs are sharded across ranks, no data-path
collective; one all_gather of the 4x4 poses (RCCL) closes the timed region.  Weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     -- the dominant kernel (fp16 MFMA coarse pass): algorithmic flops 2*N*M*D per launch
                  / average launch duration measured with HIP events on its stream, vs the dense
                  fp16 MFMA peak of MI355X_MICROARCH.md (2.5 PFLOP/s);
  cpu_baseline -- the CPU oracle (oracle/, a port: faiss and Open3D are absent) timed on this
                  box's host cores on a bounded sample of the same workload.
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


def cpu_baseline(n=N_SCAN, m=N_MAP, d=DIM, iters=RANSAC_ITERS):
    """Reference-CPU-path stand-in (kind 'port'): the oracle's restatement -- fp32 BLAS Q.B^T + exact
    fp64 decision, threshold, OpenMP RANSAC -- on a bounded sample, extrapolated linearly."""
    import numpy as np
    from oracle import oracle as orc
    from vfmreg import synth

    # 1) probe on a small sample to size the run: the whole registration is timed when it fits in
    #    ~40 s of host time, otherwise a bounded sample is extrapolated linearly (stated in `sample`).
    rows = 512
    p = synth.make_pair(n, m, d, seed=42)
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
    return {
        "value": 1.0 / total, "unit": "registrations/s", "cores": orc.num_threads(), "kind": "port",
        "sample": (f"CPU oracle (numpy BLAS fp32 Q.B^T prefilter + C/OpenMP fp64 decision and RANSAC) on {what}: "
                   f"map renorm {m}x{d} {t_norm_map:.2f}s, search of {rows_used}/{n} scan rows vs the full map "
                   f"{t_match:.2f}s, RANSAC {it_used}/{iters} iterations over {len(corres)} correspondences "
                   f"{t_ransac:.2f}s -> {total:.1f}s per registration; pose err vs planted "
                   f"{float(np.linalg.norm(res.transformation - p['T_gt'])):.4f}"),
        "host_cpu_count": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n", type=int, default=N_SCAN)
    ap.add_argument("--m", type=int, default=N_MAP)
    ap.add_argument("--iters", type=int, default=RANSAC_ITERS)
    ap.add_argument("--streams", type=int, default=2,
                    help="2: RANSAC of pair i overlaps the matching of pair i+1 on a second HIP stream; 1: serial")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from vfmreg import _lib, synth
    from vfmreg import dist as vdist
    from vfmreg.pipeline import RegistrationPipeline

    rank, world = vdist.init_from_env(backend="nccl", device=dev)  # "nccl" == RCCL on ROCm

    lib = _lib.load()
    if os.environ.get("VFM_VARIANT"):  # A/B runs (tools/r02_prof.sh): 4 = dense per-chunk records + select kernel
        lib.vfm_debug_set_coarse_variant(int(os.environ["VFM_VARIANT"]))
    n, m, d = args.n, args.m, DIM
    # two resident scene pairs per rank, alternated; pair p uses seed 42 + p (global pair id)
    pairs = [synth.make_pair_device(n, m, d, seed=42 + rank * 2 + j, device=dev) for j in range(2)]
    # --streams 2 (default): pipeline over independent scene pairs (BASELINE config C4: "one per stream"):
    # the MFMA coarse pass of pair i+1 runs on the main stream while the operand preparation of pair i+2 and
    # the select / exact decision / RANSAC of pair i run on two side streams (vfmreg/pipeline.py).  Coarse
    # passes never overlap each other, so the HIP-event duration of the coarse kernel stays a per-launch figure.
    S = 2 if args.streams >= 2 else 1
    pipe = RegistrationPipeline(n, m, d, n_iter=args.iters, device=dev, overlap_ransac=(S == 2))

    # (a high-priority matching stream was tried: no measurable difference)
    match_stream = torch.cuda.current_stream()

    torch.cuda.synchronize()
    inputs_ready = torch.cuda.Event()  # the resident scene pairs are complete from here on
    inputs_ready.record(match_stream)

    def step(i):
        p = pairs[i % 2]
        with torch.cuda.stream(match_stream):
            return pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], want_mask=True,
                                 inputs_ready=inputs_ready if S == 2 else None)

    # untimed warm-up of the complete step, including the sharding / gather path
    for i in range(max(args.warmup, 1)):
        step(i)
    with torch.cuda.stream(match_stream):
        pipe.synchronize()
    torch.cuda.current_stream().wait_stream(match_stream)
    vdist.gather_poses(torch.zeros((1, 4, 4), dtype=torch.float64, device=dev), torch.zeros(1, dtype=torch.int64, device=dev),
                       world, rank, world)
    torch.cuda.synchronize()

    events = []
    for _ in range(args.steps):
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
        events.append((a, b))
    num_pairs = world * args.steps  # global scene-pair ids; pair p runs on rank p mod world (weak scaling)
    local_i = [0]

    res_T = torch.empty((args.steps, 4, 4), dtype=torch.float64, device=dev)
    res_c = torch.empty((args.steps, 1), dtype=torch.int64, device=dev)

    def register_pair(p):
        i = local_i[0]
        local_i[0] += 1
        lib.vfm_prof_arm(events[i][0], events[i][1])
        out = step(i)
        with torch.cuda.stream(out["result_stream"]):  # snapshot the result on the producing stream
            res_T[i].copy_(out["T"])
            res_c[i].copy_(out["count"])
        return res_T[i], res_c[i]

    grouped = dist.is_available() and dist.is_initialized()  # launched through torch.distributed.run
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # every rank registers its pairs (no data-path collective), then ONE all_gather of the poses
    ids = vdist.shard_pairs(num_pairs, rank, world)
    for p_id in ids:
        register_pair(p_id)
    with torch.cuda.stream(match_stream):
        pipe.synchronize()
    torch.cuda.current_stream().wait_stream(match_stream)
    all_poses, all_counts = vdist.gather_poses(res_T, res_c.reshape(-1), num_pairs, rank, world)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    ms = C.c_float()
    durs = []
    for a, b in events:
        _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
        durs.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
    coarse_ms = sum(durs) / len(durs)

    # the same kernel without the concurrent RANSAC stream (information only; not part of `value`)
    iso = []
    if S == 2:
        pipe1 = RegistrationPipeline(n, m, d, n_iter=args.iters, device=dev, overlap_ransac=False)
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(lib.vfm_prof_events_create(C.byref(a), C.byref(b)))
        for i in range(6):
            pr = pairs[i % 2]
            lib.vfm_prof_arm(a, b)
            pipe1.register(pr["q_desc"], pr["q_xyz"], pr["b_desc"], pr["b_xyz"])
            _lib.check(lib.vfm_prof_elapsed_ms(a, b, C.byref(ms)))
            if i:
                iso.append(ms.value)
        lib.vfm_prof_events_destroy(a, b)
        del pipe1
    iso_ms = (sum(iso) / len(iso)) if iso else coarse_ms

    # sanity of the timed work: every pose must recover the planted transform
    import numpy as np
    mine = vdist.shard_pairs(num_pairs, rank, world)
    errs = [float(np.linalg.norm(all_poses[p].cpu().numpy() - pairs[i % 2]["T_gt"])) for i, p in enumerate(mine)]
    ncorr = int(all_counts[mine[-1]].item())

    if rank == 0:
        flops = 2.0 * n * m * d
        achieved = flops / (coarse_ms * 1e-3) / 1e12
        traffic = None  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
        pmc = ROOT / "profiles" / "r01_pmc_match_coarse.json"
        if pmc.exists() and (n, m, d) == (N_SCAN, N_MAP, DIM):
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        line = {
            "metric": "registrations/sec (20k<->200k pts, 384-D)", "value": world * args.steps / elapsed,
            "unit": "registrations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16 coarse pass (MFMA) + f64 exact decision / f64 RANSAC",
            "data": "synthetic",
            "config": {"workload": f"C2: {n}-pt scan vs {m}-pt map, {d}-D descriptors precomputed and resident in "
                                   f"HBM, {args.iters} RANSAC iterations, cosine >= 0.8; map renormalised every step",
                       "registrations_per_gpu": args.steps, "resident_scene_pairs_per_gpu": 2,
                       "parallelism": f"{world} GPU shard(s) x {S} stream(s), independent scene pairs",
                       "collective": (f"one all_gather_into_tensor of the poses ({dist.get_backend()} = RCCL)" if grouped
                                      else "none (single process, no launcher)"),
                       "correspondences_last_step": ncorr, "max_pose_err_vs_planted": max(errs)},
            "roofline": {"bound": "mfma", "kernel": "match_coarse_pipe_kernel<24> (fp16 32x32x16 MFMA, fused top-2 epilogue)",
                         "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_match_coarse.json (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes)",
                         "flops_per_launch": flops, "avg_launch_ms": coarse_ms,
                         "single_stream": {"avg_launch_ms": iso_ms, "achieved": flops / (iso_ms * 1e-3) / 1e12,
                                           "frac": flops / (iso_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                           "note": "same kernel without the RANSAC of the previous pair running beside it"}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n, m, d, args.iters)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
