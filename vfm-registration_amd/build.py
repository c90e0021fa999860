"""Build libvfmreg_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python vfm-registration_amd/build.py [--force]

hipcc cross-compiles without a GPU.  The .so lands in vfm-registration_amd/vfmreg/lib/ so it
travels with the source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT_DIR = HERE / "vfmreg" / "lib"
SO = OUT_DIR / "libvfmreg_hip.so"
OBJ_DIR = HERE / "build"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the fp64 / fp32 parity kernels must not fuse a*b+c
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]
# per-file additions.  vit.hip: MFMA results in architectural registers -- vit_qkv_attention_kernel runs one wave per SIMD with 512 registers, and
# with the accumulators in the accumulation file every value the softmax and the epilogues touch was moved across first (936 v_accvgpr_read
# in 5 900 instructions); the kernels that stay under 256 registers do not change
FILE_FLAGS = {"vit.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
# ... and vit.hip once more for its second part (vit_mlp_kernel: 192 accumulators in the accumulation file), without that option
EXTRA_OBJECTS = [("vit.hip", ["-DVFM_VIT_PART=1"], "vit_mlp.hip.o")]
SOURCES = ["error.cpp", "config.cpp", "match_api.hip", "match_prep.hip", "match_coarse_f16.hip", "match_coarse_i8.hip", "match_coarse_mx6.hip", "match_finish.hip",
           "match_l2.hip", "ransac.hip", "project.hip", "vit.hip", "icp.hip", "voxel.hip"]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    OUT_DIR.mkdir(parents=True, exist_ok=True)
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [HERE.parent / "include" / "vfmreg.h"]
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objs = [OBJ_DIR / (s.name + ".o") for s in srcs]
    jobs = [(s, o, FILE_FLAGS.get(s.name, [])) for s, o in zip(srcs, objs)]
    for name, flags, objname in EXTRA_OBJECTS:
        if (CSRC / name).exists():
            jobs.append((CSRC / name, OBJ_DIR / objname, flags))
            objs.append(OBJ_DIR / objname)

    def compile_one(job):
        src, obj, extra = job
        if not force and not _stale(obj, [src] + headers):
            return
        cmd = [HIPCC] + FLAGS + extra + (["-x", "hip"] if src.suffix == ".cpp" else []) + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(compile_one, jobs))
    if force or _stale(SO, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(SO)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
