"""DINOv2 ViT-S/14 (+ FeatUp ChannelNorm) weights and forward wrapper (row A1).

The reference obtains the model with ``torch.hub.load("mhamilton723/FeatUp", "dinov2",
use_norm=True)`` (image_features.py:39-42) -- no network here, so weights come either from a checkpoint's
state dict (``load_state_dict``: facebookresearch/dinov2, FeatUp-wrapper or transformers key layouts; the FeatUp
ChannelNorm becomes ``channel_norm.{weight,bias}``) or from ``random_weights`` (seeded, exact ViT-S/14 shapes).

Host-side work done once per (weights, input resolution): bicubic interpolation of the 37x37
position embedding to the 16 x pw patch grid (dinov2 ``interpolate_pos_encoding``; depends only on
the shape) and re-tiling of the linear weights into the fp16 MFMA fragment layout the HIP kernels
read (see csrc/vit.hip).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops

PATCH = 14
PATCH_H = 16  # image_features.py:35


def vit_s14_shapes(dim: int = 384, depth: int = 12, mlp: int = 1536, n_pos: int = 37 * 37 + 1) -> Dict[str, tuple]:
    s = {"patch_embed.proj.weight": (dim, 3, PATCH, PATCH), "patch_embed.proj.bias": (dim,),
         "cls_token": (1, 1, dim), "pos_embed": (1, n_pos, dim), "norm.weight": (dim,), "norm.bias": (dim,),
         "channel_norm.weight": (dim,), "channel_norm.bias": (dim,)}
    for i in range(depth):
        p = f"blocks.{i}."
        s.update({p + "norm1.weight": (dim,), p + "norm1.bias": (dim,), p + "attn.qkv.weight": (3 * dim, dim),
                  p + "attn.qkv.bias": (3 * dim,), p + "attn.proj.weight": (dim, dim), p + "attn.proj.bias": (dim,),
                  p + "ls1.gamma": (dim,), p + "norm2.weight": (dim,), p + "norm2.bias": (dim,),
                  p + "mlp.fc1.weight": (mlp, dim), p + "mlp.fc1.bias": (mlp,), p + "mlp.fc2.weight": (dim, mlp),
                  p + "mlp.fc2.bias": (dim,), p + "ls2.gamma": (dim,)})
    return s


def random_weights(seed: int = 0, dim: int = 384, depth: int = 12, mlp: int = 1536) -> Dict[str, np.ndarray]:
    """Seeded stand-in for the pretrained checkpoint (no network): every branch contributes at O(1)."""
    rng = np.random.default_rng(seed)
    w = {}
    for k, shp in vit_s14_shapes(dim, depth, mlp).items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k in ("norm.weight", "channel_norm.weight"):
            v = 1.0 + 0.1 * rng.standard_normal(shp)
        elif k.endswith(".gamma"):
            v = rng.uniform(0.05, 0.5, shp)
        elif k.endswith(".bias"):
            v = 0.05 * rng.standard_normal(shp)
        elif k in ("cls_token", "pos_embed"):
            v = 0.2 * rng.standard_normal(shp)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = rng.standard_normal(shp) / math.sqrt(fan_in)
        w[k] = v.astype(np.float32)
    return w


def dinov2_like_weights(seed: int = 0, dim: int = 384, depth: int = 12, mlp: int = 1536, outlier_channels: int = 6,
                        outlier_scale=(50.0, 200.0), layerscale_range=(1e-4, 1.0)) -> Dict[str, np.ndarray]:
    """``random_weights`` with the two statistics of a TRAINED DINOv2 that a seeded initialisation lacks (no checkpoint can be
    fetched here): a handful of residual-stream channels that carry activations 50-200x the others in every token and through
    every block (planted in the patch embedding's bias and the position embedding, i.e. on the residual path itself), and
    LayerScale gammas spread log-uniformly over four decades.  The LayerNorms in front of QKV / fc1 / the output then normalise rows
    whose variance is dominated by those channels: the regime in which an fp16 operand of the un-normalised stream
    (csrc/vit.hip folds the LayerNorm into the consuming GEMM) has to hold up.  Returns (weights, outlier channel indices)."""
    rng = np.random.default_rng(seed + 7919)
    w = random_weights(seed, dim, depth, mlp)
    ch = np.sort(rng.choice(dim, size=outlier_channels, replace=False))
    amp = rng.uniform(outlier_scale[0], outlier_scale[1], outlier_channels) * rng.choice([-1.0, 1.0], outlier_channels)
    w["patch_embed.proj.bias"][ch] += amp.astype(np.float32)
    w["cls_token"][0, 0, ch] += amp.astype(np.float32)                 # the cls token has no patch embedding: same channels, same size
    w["pos_embed"][0, :, ch] += (0.05 * amp[:, None] * rng.standard_normal((outlier_channels, w["pos_embed"].shape[1]))).astype(np.float32)
    lo, hi = math.log(layerscale_range[0]), math.log(layerscale_range[1])
    for k in w:
        if k.endswith(".gamma"):
            w[k] = np.exp(rng.uniform(lo, hi, w[k].shape)).astype(np.float32)
    return w, ch


def load_state_dict(state_dict, channel_norm: str = "identity") -> Dict[str, np.ndarray]:
    """Weights dict for ``ViTS14`` / ``ImageFeatureGenerator(weights=...)`` from a checkpoint's state dict
    (torch tensors or numpy arrays).  Accepted key layouts:

    * facebookresearch/dinov2 (``torch.hub.load('facebookresearch/dinov2', 'dinov2_vits14').state_dict()``):
      ``cls_token, pos_embed, patch_embed.proj.*, blocks.<i>.{norm1, attn.qkv, attn.proj, ls1.gamma, norm2, mlp.fc1,
      mlp.fc2, ls2.gamma}, norm.*`` -- also with chunked blocks (``blocks.<chunk>.<i>.``); ``mask_token`` is ignored;
    * the FeatUp wrapper the reference loads (image_features.py:39-42): ``model.0.model.<dinov2 key>`` for the
      featurizer and ``model.1.norm.{weight,bias}`` for ``ChannelNorm(384)`` (``use_norm=True``); ``upsampler.*`` (the JBU
      stack, unused with ``use_featup=False``) is ignored;
    * Hugging Face ``transformers.Dinov2Model``: ``embeddings.*, encoder.layer.<i>.*, layernorm.*`` (separate
      query / key / value projections are concatenated into the fused qkv).

    A plain DINOv2 checkpoint has no ChannelNorm: ``channel_norm='identity'`` then uses LayerNorm's initial affine
    (weight 1, bias 0), ``'require'`` raises instead.  Register tokens (dinov2 ``*_reg`` models) are not supported."""
    sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dict.items()}
    sd = {k: v.astype(np.float32) for k, v in sd.items() if v.dtype.kind == "f"}
    out: Dict[str, np.ndarray] = {}
    if any(k.startswith("model.0.model.") for k in sd):  # FeatUp UpsampledBackbone
        for k, v in sd.items():
            if k.startswith("model.0.model."):
                out[k[len("model.0.model."):]] = v
            elif k in ("model.1.norm.weight", "model.1.norm.bias"):
                out["channel_norm." + k.rsplit(".", 1)[1]] = v
    elif any(k.startswith("embeddings.") or k.startswith("dinov2.embeddings.") for k in sd):  # transformers
        sd = {(k[len("dinov2."):] if k.startswith("dinov2.") else k): v for k, v in sd.items()}
        ren = {"embeddings.cls_token": "cls_token", "embeddings.position_embeddings": "pos_embed",
               "embeddings.patch_embeddings.projection.weight": "patch_embed.proj.weight",
               "embeddings.patch_embeddings.projection.bias": "patch_embed.proj.bias",
               "layernorm.weight": "norm.weight", "layernorm.bias": "norm.bias"}
        for a, b in ren.items():
            if a in sd:
                out[b] = sd[a]
        i = 0
        while f"encoder.layer.{i}.norm1.weight" in sd:
            q, p = f"encoder.layer.{i}.", f"blocks.{i}."
            for wb in ("weight", "bias"):
                out[p + "attn.qkv." + wb] = np.concatenate([sd[q + f"attention.attention.{n}.{wb}"]
                                                            for n in ("query", "key", "value")], axis=0)
                out[p + "attn.proj." + wb] = sd[q + "attention.output.dense." + wb]
                for n in ("norm1", "norm2", "mlp.fc1", "mlp.fc2"):
                    out[p + n + "." + wb] = sd[q + n + "." + wb]
            out[p + "ls1.gamma"] = sd[q + "layer_scale1.lambda1"]
            out[p + "ls2.gamma"] = sd[q + "layer_scale2.lambda1"]
            i += 1
        for k in ("channel_norm.weight", "channel_norm.bias"):
            if k in sd:
                out[k] = sd[k]
    else:  # facebookresearch/dinov2
        import re
        for k, v in sd.items():
            out[re.sub(r"^blocks\.\d+\.(\d+)\.", r"blocks.\1.", k)] = v
    if "register_tokens" in out or any("register_tokens" in k for k in sd):
        raise NotImplementedError("DINOv2 checkpoints with register tokens")
    out.pop("mask_token", None)
    if "channel_norm.weight" not in out:
        if channel_norm == "require":
            raise KeyError("the state dict has no ChannelNorm (FeatUp `model.1.norm.*` / `channel_norm.*`)")
        dim = out["patch_embed.proj.weight"].shape[0]
        out["channel_norm.weight"] = np.ones(dim, np.float32)
        out["channel_norm.bias"] = np.zeros(dim, np.float32)
    dim = out["patch_embed.proj.weight"].shape[0]
    depth = 0
    while f"blocks.{depth}.norm1.weight" in out:
        depth += 1
    mlp = out["blocks.0.mlp.fc1.weight"].shape[0]
    want = vit_s14_shapes(dim, depth, mlp, n_pos=out["pos_embed"].shape[1])
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError(f"state dict lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
    for k, shp in want.items():
        if tuple(out[k].shape) != tuple(shp):
            raise ValueError(f"{k}: shape {tuple(out[k].shape)}, expected {tuple(shp)}")
    if out["patch_embed.proj.weight"].shape[2:] != (PATCH, PATCH) or dim % 64:
        raise ValueError("not a patch-14 ViT with 64-wide heads")
    return {k: np.ascontiguousarray(out[k]) for k in want}


def interpolate_pos_embed(pos_embed: np.ndarray, h: int, w: int) -> np.ndarray:
    """dinov2 interpolate_pos_encoding (bicubic, +0.1 offset trick) -> [1 + h*w, dim] fp32."""
    import torch.nn.functional as F
    pe = torch.as_tensor(pos_embed, dtype=torch.float32)
    n = pe.shape[1] - 1
    m = int(round(math.sqrt(n)))
    dim = pe.shape[-1]
    if h == m and w == m:
        return pe[0].numpy()
    patch = F.interpolate(pe[:, 1:].reshape(1, m, m, dim).permute(0, 3, 1, 2),
                          scale_factor=((h + 0.1) / m, (w + 0.1) / m), mode="bicubic", align_corners=False)
    assert patch.shape[-2:] == (h, w)
    patch = patch.permute(0, 2, 3, 1).reshape(h * w, dim)
    return torch.cat([pe[0, :1], patch], 0).numpy()


def to_frag_f16(W: np.ndarray, k_multiple: int = 16) -> np.ndarray:
    """[N, K] fp32 -> fp16 fragment tiles [N/32][K/16][2][32][8] (N, K zero-padded to 32 / ``k_multiple``)."""
    n, k = W.shape
    npad, kpad = -(-n // 32) * 32, -(-k // k_multiple) * k_multiple
    Wp = np.zeros((npad, kpad), dtype=np.float16)
    Wp[:n, :k] = W.astype(np.float16)
    return np.ascontiguousarray(Wp.reshape(npad // 32, 32, kpad // 16, 2, 8).transpose(0, 2, 3, 1, 4))


class ViTS14:
    """Device-resident DINOv2 ViT-S/14 + ChannelNorm for one input resolution."""

    def __init__(self, weights: Dict[str, np.ndarray], img_h: int, img_w: int, device="cuda"):
        lib = _lib.load()
        self.device = torch.device(device)
        self.img_h, self.img_w = img_h, img_w
        scale = (PATCH * PATCH_H) / img_h            # image_features.py:68
        self.patch_w = int(scale * img_w / PATCH)    # image_features.py:69
        dim = weights["patch_embed.proj.weight"].shape[0]
        depth = 0
        while f"blocks.{depth}.norm1.weight" in weights:
            depth += 1
        mlp = weights["blocks.0.mlp.fc1.weight"].shape[0]
        self.cfg = _lib.VitConfig(dim, depth, dim // 64, mlp, PATCH, PATCH_H, self.patch_w)
        self.dim = dim
        nseg = 3 + 12 * depth + 4
        offs = (C.c_int64 * nseg)()
        sizes = (C.c_int64 * nseg)()
        cnt = lib.vfm_vit_weights_layout(C.byref(self.cfg), offs, sizes, nseg)
        assert cnt == nseg
        total = lib.vfm_vit_weights_bytes(C.byref(self.cfg))
        blob = np.zeros(total, dtype=np.uint8)

        def put(i, arr):
            raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
            assert raw.size == sizes[i], (i, raw.size, sizes[i])
            blob[offs[i]:offs[i] + raw.size] = raw

        g = lambda k: np.asarray(weights[k], dtype=np.float32)
        pos = interpolate_pos_embed(g("pos_embed"), PATCH_H, self.patch_w)
        cls_pos = pos.copy()
        cls_pos[0] += g("cls_token").reshape(-1)
        put(0, to_frag_f16(g("patch_embed.proj.weight").reshape(dim, -1), 32))   # (588 -> 608 columns: whole stages of two k-steps, csrc/vit.hip)
        put(1, g("patch_embed.proj.bias"))
        put(2, cls_pos.astype(np.float32))
        i = 3
        def folded(W, b, gamma, beta):
            """LayerNorm (gamma, beta) folded into the linear layer (W, b) behind it (csrc/vit.hip, Seg): the fp16 fragment
            tiles of W diag(gamma), b + W beta, and the row sums of the weight AS ROUNDED (what the GEMM multiplies)."""
            Wf = (W.astype(np.float64) * gamma.astype(np.float64)[None, :]).astype(np.float32)
            Wh = Wf.astype(np.float16)
            return (to_frag_f16(Wf), (b.astype(np.float64) + W.astype(np.float64) @ beta.astype(np.float64)).astype(np.float32),
                    Wh.astype(np.float64).sum(1).astype(np.float32))
        for l in range(depth):
            p = f"blocks.{l}."
            qw, qb, qc = folded(g(p + "attn.qkv.weight"), g(p + "attn.qkv.bias"), g(p + "norm1.weight"), g(p + "norm1.bias"))
            fw, fb, fc = folded(g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"), g(p + "norm2.weight"), g(p + "norm2.bias"))
            for arr in (qw, qb, qc, to_frag_f16(g(p + "attn.proj.weight")), g(p + "attn.proj.bias"), g(p + "ls1.gamma"),
                        fw, fb, fc, to_frag_f16(g(p + "mlp.fc2.weight")), g(p + "mlp.fc2.bias"), g(p + "ls2.gamma")):
                put(i, arr)
                i += 1
        for name in ("norm.weight", "norm.bias", "channel_norm.weight", "channel_norm.bias"):
            put(i, g(name))
            i += 1
        self.blob = torch.from_numpy(blob).to(self.device)
        self._ws = {}
        self._side = None   # two side streams for batches from SPLIT_FROM images on
        self._pool = None   # ... and the helper thread that enqueues the second half

    # OPT-IN (round 5): batches from SPLIT_FROM images on run as two half-batches on two side streams, the second half enqueued by a helper
    # thread.  Every kernel of a forward ends in a tail -- its workgroups start together, and as they finish the compute units run three,
    # then two, then one of them, the last at a quarter of the matrix pipes -- and the next kernel cannot start under it;
    # two independent half-batches fill each other's tails.  tools/ab_vit_two_halves.py and profiles/r05_ab_vit_two_streams.txt, one
    # forward -> two halves, forwards back to back: 66 images 2.39 -> 2.25 ms, 72: 2.54 -> 2.36, 90: 2.96 -> 2.89, 96: 3.33 -> 3.09,
    # 120: 4.0 -> 3.85; a single synchronised forward gains half of that; below 64 images the halves fall under the sizes the batch kernels
    # are chosen for and lose.  Identical outputs (the images are independent).  Off by default (0): which hardware queue a side stream
    # lands on depends on what the process created before (pipeline.py, _side_streams), and a figure that moves with that is not a default;
    # a caller that walks a sequence in batches sets ViTS14.SPLIT_FROM = 64.
    SPLIT_FROM = 0

    def forward(self, images: torch.Tensor, out: Optional[torch.Tensor] = None, _slot: int = 0) -> torch.Tensor:
        """images: [B, H, W, 3] uint8 on the device -> [B, 16, pw, dim] fp32 patch features (written into ``out`` when given:
        a consumer that holds pointers into it -- ops.LiftPlan -- then needs no re-marshalling)."""
        ops._chk(images, torch.uint8, "images")
        B, H, W, _ = images.shape
        if (H, W) != (self.img_h, self.img_w):
            raise ValueError("Invalid shape")
        lib = _lib.load()
        if out is None:
            out = torch.empty((B, PATCH_H, self.patch_w, self.dim), dtype=torch.float32, device=self.device)
        else:
            ops._chk(out, torch.float32, "out")
            if tuple(out.shape) != (B, PATCH_H, self.patch_w, self.dim):
                raise ValueError("Invalid shape")
        if self.SPLIT_FROM and B >= self.SPLIT_FROM and _slot == 0:
            if self._side is None:
                self._side = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
            main = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(main)
            h = (B + 1) // 2

            cfg = _lib.current()   # the caller's kernel policy (vfm_config_t, bound per thread): the helper thread binds the same one

            def half(k, lo, hi):
                self._side[k].wait_event(ready)
                with _lib.using(cfg), torch.cuda.stream(self._side[k]):
                    self.forward(images[lo:hi], out[lo:hi], _slot=k + 1)

            # the second half is enqueued by a helper thread while this one enqueues the first (the C call releases the GIL): enqueued one
            # after the other the second half started ~0.3 ms -- 65 launches -- behind the first, and a single forward lost what the
            # overlap gained
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(1)
            fut = self._pool.submit(half, 1, h, B)
            try:
                half(0, 0, h)
            finally:
                # (whatever the first half did, the helper is joined and the caller's stream waits for both side streams --
                #  no half-enqueued forward is left running on buffers the caller believes free)
                try:
                    fut.result()
                finally:
                    for st in self._side:
                        main.wait_stream(st)
            return out
        key = (B, _slot)
        if key not in self._ws:
            self._ws[key] = torch.empty(lib.vfm_vit_workspace_bytes(C.byref(self.cfg), B), dtype=torch.uint8,
                                        device=self.device)
        _lib.check(lib.vfm_vit_forward(C.byref(self.cfg), self.blob.data_ptr(), images.data_ptr(), B, H, W,
                                       out.data_ptr(), self._ws[key].data_ptr(), self._ws[key].numel(), ops._stream()),
                   "vit_forward")
        return out
