"""ViT-S/14 on the 6 images of one scan: one chain of 63 launches over all images against the images split over several streams
(each part its own chain of 63 launches over fewer tokens, the chains side by side).   python tools/ab_vit_split.py"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vfm-registration_amd"))
from vfmreg import vit as V  # noqa: E402

dev = torch.device("cuda")
rng = np.random.default_rng(0)
B, H, W = 6, 1200, 1600
imgs = torch.from_numpy(rng.integers(1, 255, (B, H, W, 3), dtype=np.uint8)).to(dev)
w = V.random_weights(0)
models = [V.ViTS14(w, H, W, device=dev) for _ in range(6)]
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
ref = models[0].forward(imgs).clone()
torch.cuda.synchronize()
for parts in (1, 2, 3, 6, 1, 2, 3, 6):
    per = B // parts
    out = torch.empty_like(ref)
    def run():
        if parts == 1:
            models[0].forward(imgs, out)
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(main)
        for p in range(parts):
            s = streams[p]
            s.wait_event(ev)
            with torch.cuda.stream(s):
                models[p].forward(imgs[p * per:(p + 1) * per], out[p * per:(p + 1) * per])
        for p in range(parts):
            main.wait_stream(streams[p])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    eager = e0.elapsed_time(e1) / 20
    same = bool(torch.equal(out, ref))
    # the same as a graph: the host's launch pace (63 launches per chain at ~5 us each) drops out
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(device=dev)
    cs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cs):
        out.zero_()
        with torch.cuda.graph(g, stream=cs):
            run()
    torch.cuda.current_stream().wait_stream(cs)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{parts} chain(s) of {per} image(s): {eager:.3f} ms per scan launched, {e0.elapsed_time(e1) / 20:.3f} ms as a graph; identical to one chain: {same} / {bool(torch.equal(out, ref))}", flush=True)
