#!/usr/bin/env python
"""C2-size registration on a NEAR-DUPLICATE-RICH map: map descriptors are lifted the way A1->A3
lifts them -- bilinear interpolation of per-image 16 x 21 patch grids, several overlapping clouds -- instead of isolated
random unit vectors.  Prints the candidate-entries-per-query histogram of match_select_kernel, the number of queries the
fp32 refinement handled, the all-pairs fallbacks, and registrations/s next to the random-descriptor figure.

    python tools/time_neardup.py [--steps 30] [--out profiles/r02_neardup.json]
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, synth  # noqa: E402
from vfmreg.pipeline import RegistrationPipeline  # noqa: E402


lifted_map = synth.lifted_map   # moved into the package (bench.py's C2_lifted uses it too)


def run(pipe, p, steps, lib, n, m):
    ev = torch.cuda.Event()
    ev.record()
    for i in range(8):   # warm-up, one registration at a time: the "auto" policy settles (probe -> records by feedback)
        pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
        pipe.synchronize()
        torch.cuda.synchronize()
        pipe._poll_feedback()
    t0 = time.perf_counter()
    for i in range(steps):
        out = pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
    pipe.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lib.vfm_debug_set_match_stats(1)  # one more registration with the counters on (they cost atomics: not timed)
    pipe.register(p["q_desc"], p["q_xyz"], p["b_desc"], p["b_xyz"], inputs_ready=ev)
    pipe.synchronize()
    torch.cuda.synchronize()
    lib.vfm_debug_set_match_stats(0)
    st = (C.c_int32 * 64)()
    r = pipe.sets[(pipe._step - 1) % len(pipe.sets)]
    _lib.check(lib.vfm_debug_match_stats(r.sws.data_ptr(), n, m, C.cast(st, C.c_void_p)))
    st = list(st)
    return dt, st, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--out", default="")
    ap.add_argument("--rows", type=int, default=0, help="only the first so many maps")
    ap.add_argument("--modes", default="auto,int8-half,int8,mx6,int8-top2,fp16")
    a = ap.parse_args()
    lib = _lib.load()
    n, m, d = 20000, 200000, 384
    dev = torch.device("cuda")
    res = {}
    base = synth.make_pair_device(n, m, d, seed=42)
    cases = {"random (D.2)": None,
             "lifted: 10 clouds x 6 cams, independent views": dict(clouds=10, view_noise=1.0),
             "lifted: 10 clouds x 6 cams, views share the scene (noise 0.1)": dict(clouds=10, view_noise=0.1),
             "lifted: 30 clouds x 6 cams, views share the scene (noise 0.02)": dict(clouds=30, view_noise=0.02),
             "revisited: 20k physical points seen by 10 clouds each (view noise 0.02)": dict(clouds=10, view_noise=0.02, revisit=3334),
             "revisited: 4k physical points seen by 50 clouds each (view noise 0.01)": dict(clouds=50, view_noise=0.01, revisit=667),
             "revisited: 1k physical points seen by 200 clouds each (view noise 0.01)": dict(clouds=200, view_noise=0.01, revisit=167)}
    for name, cfg in list(cases.items())[:a.rows or None]:
        p = dict(base)
        if cfg is not None:
            b = lifted_map(m, d, cfg["clouds"], 6, 16, 21, 7, cfg["view_noise"], dev, cfg.get("revisit", 0))
            g = torch.Generator(device=dev)
            g.manual_seed(1)
            pick = base["match"].clamp(min=0)
            rms = b.pow(2).mean().sqrt()
            q = b[pick] + 0.3 * rms * torch.randn((n, d), generator=g, device=dev)
            is_out = base["match"] < 0
            q = torch.where(is_out[:, None], torch.randn((n, d), generator=g, device=dev), q)
            p["b_desc"], p["q_desc"] = b.contiguous(), q.contiguous()
        ref = None
        for coarse in a.modes.split(","):  # the bench's pipeline: prepare on its own stream, two solve streams
            pipe = RegistrationPipeline(n, m, d, n_iter=50000, overlap_ransac=True, overlap_prepare=True, solve_streams=2, coarse=coarse)
            dt, st, out = run(pipe, p, a.steps, lib, n, m)
            k = int(out["count"].item())
            err = float(np.linalg.norm(out["T"].cpu().numpy() - base["T_gt"]))
            same = True
            if ref is None:
                ref = (out["T"].clone(), out["corres"][:k].clone())
            else:  # the three passes must agree on the correspondences and the pose
                same = bool(torch.equal(ref[0], out["T"]) and torch.equal(ref[1], out["corres"][:k]))
            key = name + " | " + coarse
            hist = {f"<= {1 << bnum}": st[8 + bnum] for bnum in range(16) if st[8 + bnum]}
            res[key] = dict(ms_per_registration=1e3 * dt, registrations_per_s=1.0 / dt, correspondences=k, pose_err_vs_planted=err,
                            pass_in_use=(("fp6" if ((pipe.half and getattr(pipe, "mx6_half", False)) or (not pipe.half and getattr(pipe, "mx6", False))) else "int8")
                                         if pipe.use_i8 else "fp16"),
                            records_in_use=("half-width" if (pipe.use_i8 and pipe.half) else "top-2" if (pipe.top2 or getattr(pipe, "mx6_top2", False)) else "best score"),
                            same_result_as_auto=same,
                            rescanned_chunks_per_query=(pipe.last_rescans / n) if pipe.last_rescans is not None else None,
                            half_width_probe_survivors_per_query=(pipe.last_probe / n) if getattr(pipe, "last_probe", None) is not None else None,
                            fallback_queries=st[0], refined_queries=st[1], coarse_records_per_query=st[4] / n, candidate_entries_per_query=st[2] / n,
                            rows_kept_per_refined_query=(st[3] / st[1]) if st[1] else 0.0, candidate_entry_histogram=hist)
            print(key, json.dumps(res[key]), flush=True)
            del pipe
    if a.out:
        Path(a.out).parent.mkdir(parents=True, exist_ok=True)
        Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
