"""Profiling driver: runs only the matching stage (prepare + coarse + select + rescore) of config C2
a few times, so that rocprofv3 --pmc passes focus on match_coarse_kernel.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/prof_match.py"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
import torch  # noqa: E402

import os  # noqa: E402

from vfmreg import _lib  # noqa: E402

if os.environ.get("VFM_LIB"):  # experimental build (tools/build_ablate.sh)
    _lib.LIB_PATH = ROOT / "vfm-registration_amd" / "vfmreg" / "lib" / os.environ["VFM_LIB"]
from vfmreg import ops, synth  # noqa: E402

_lib.load().vfm_debug_set_coarse_variant(int(os.environ.get("VFM_VARIANT", "0")))
_lib.load().vfm_debug_set_coarse_slices(int(os.environ.get("VFM_SLICES", "0")))

n, m, d = 20000, 200000, 384
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
p = synth.make_pair_device(n, m, d, seed=42)
for _ in range(reps):
    # the gated family (int8 pass for d = 256 / 384) with the pipeline's gate; VFM_GATE=none -> the ungated call (fp16 pass)
    g = os.environ.get("VFM_GATE", "0.8")
    idx, sim = ops.match_ip_top1(p["q_desc"], p["b_desc"], ops.FAST, gate=None if g == "none" else float(g))
torch.cuda.synchronize()
print("ok", int((idx == p["match"]).sum()), "unresolved", int((idx < 0).sum()))  # meaningless for ablated builds
