#!/usr/bin/env python
"""Phase times of vit_qkv_attention_kernel's workgroups (its trace: 100 MHz ticks at start, tokens in registers, K done, V^T done, q done,
first query tile done, end) in the last layer of a forward.  python tools/trace_vit_fused.py [nimg ...]"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import _lib, vit as V  # noqa: E402
lib = _lib.load()
rng = np.random.default_rng(0)
buf = torch.zeros((4096, 16), dtype=torch.int64, device="cuda")
ptr = buf.data_ptr()
lib.vfm_debug_set_vit_gemm(-11, C.c_int32(ptr & 0xffffffff).value)
lib.vfm_debug_set_vit_gemm(-12, C.c_int32((ptr >> 32) & 0xffffffff).value)
_lib.thread_config().set("vit_fused_qkv", 1)
_lib.thread_config().set("vit_trace_fused", 1)
model = V.ViTS14(V.random_weights(0, depth=2), 1200, 1600)
for nimg in [int(x) for x in (sys.argv[1:] or ["6", "42", "84", "90"])]:
    imgs = torch.from_numpy(rng.integers(1, 255, (nimg, 1200, 1600, 3), dtype=np.uint8)).cuda()
    for _ in range(3):
        buf.zero_()
        model.forward(imgs)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    rel = (t[:, :7] - t0) / 100.0
    ph = np.diff(rel, axis=1)
    names = ["tokens", "K", "V^T", "q", "att tile 0", "att rest"]
    sub = (t[:, [7, 8, 9, 5]] - t[:, [4, 7, 8, 9]]) / 100.0
    first = rel[:, 0] < 5.0
    print(f"{nimg} images, {len(t)} workgroups, kernel {rel[:, 6].max():.1f} us; workgroups started in the first 5 us: {int(first.sum())}; "
          f"duration median {np.median(rel[:, 6] - rel[:, 0]):.1f} us (first round {np.median((rel[:, 6] - rel[:, 0])[first]):.1f}, later {np.median((rel[:, 6] - rel[:, 0])[~first]) if (~first).any() else 0:.1f})")
    v1 = (t[:, [10, 11, 3]] - t[:, [2, 10, 11]]) / 100.0   # tile v1: from the end of v0 (tr[2] is the end of k1: v0 + wait), k-loop, epilogue
    q0 = (t[:, [13, 14, 15]] - t[:, [3, 13, 14]]) / 100.0
    print(f"    tile v1: [v0 + barrier] {np.median(v1[:, 0]):.2f}, k-loop {np.median(v1[:, 1]):.2f}, epilogue {np.median(v1[:, 2]):.2f} us;   "
          f"tile q0: barrier {np.median(q0[:, 0]):.2f}, k-loop {np.median(q0[:, 1]):.2f}, epilogue {np.median(q0[:, 2]):.2f} us")
    for k, nm in enumerate(names):
        if k == 5:
            for j, sn in enumerate(["scores", "softmax", "P V", "stores"]):
                print(f"        tile 0 {sn:8s} median {np.median(sub[:, j]):6.2f} us   max {sub[:, j].max():6.2f}")
        print(f"    {nm:12s} median {np.median(ph[:, k]):6.2f} us   first round {np.median(ph[first, k]):6.2f}   later {np.median(ph[~first, k]) if (~first).any() else 0:6.2f}   max {ph[:, k].max():6.2f}")
