"""Mirror of vfm_reg.image_features.ImageFeatureGenerator (image_features.py:23-117) for the
configuration the registration path uses: ``ImageFeatureGenerator('dinov2', use_featup=False)``
(registration_node.py:57, prepare_scenes.py:121).

``get_image_features(image, upsample=False)`` returns the 16 x pw x 384 patch features (HWC fp32
numpy) computed by the HIP ViT (csrc/vit.hip); ``upsample=True`` reproduces the bilinear
``F.interpolate`` to H x W (image_features.py:104-108) with the fused gather kernel -- but
``create_descriptors`` never materialises that 2.9 GB tensor: it calls ``patch_features_device``
and lets the gather kernel interpolate per projected point.

No network: pass ``weights`` -- a checkpoint path or a state dict in facebookresearch/dinov2, FeatUp-wrapper or
transformers naming (``vit.load_state_dict``); without it a seeded random ViT-S/14 is used and a warning is printed.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from . import vit as V


class ImageFeatureGenerator:
    def __init__(self, foundation_model: str, use_featup: bool = True, weights: Optional[Dict[str, np.ndarray]] = None,
                 device="cuda"):
        self.foundation_model_name = foundation_model
        self.use_featup = use_featup
        self.device = device
        self.patch_h = 16   # image_features.py:35
        self.patch_w = None
        if foundation_model == "dinov2":
            self.patch_size, self.feature_size = 14, 384
        elif foundation_model == "maskclip":
            raise NotImplementedError("maskclip is an ablation model outside the hot path (SURVEY.md section 2 row 1)")
        else:
            raise ValueError(f"Unsupported foundation model: {foundation_model}")  # image_features.py:54
        if use_featup:
            raise NotImplementedError("the FeatUp JBU upsampler is not on the hot path: the reference builds the "
                                      "generator with use_featup=False (registration_node.py:57)")
        if isinstance(weights, (str, Path)):  # a checkpoint file: dinov2 / FeatUp / transformers key layouts
            sd = torch.load(weights, map_location="cpu", weights_only=True)
            weights = V.load_state_dict(sd.get("state_dict", sd) if isinstance(sd, dict) else sd)
        elif weights is not None and "patch_embed.proj.weight" not in weights:
            weights = V.load_state_dict(weights)  # a raw state dict in one of the supported layouts
        if weights is None:
            print("[WARNING] no DINOv2 weights given: using seeded random ViT-S/14 weights")
            weights = V.random_weights(seed=0)
        self.weights = weights
        self.feature_size = weights["patch_embed.proj.weight"].shape[0]
        self._models: Dict[tuple, V.ViTS14] = {}
        self.image_shape = (-1, -1)

    def create_transform_(self, img_h, img_w) -> None:  # image_features.py:67-77
        scale = (self.patch_size * self.patch_h) / img_h
        self.patch_w = int(scale * img_w / self.patch_size)
        self.image_shape = (img_h, img_w)
        if (img_h, img_w) not in self._models:
            self._models[(img_h, img_w)] = V.ViTS14(self.weights, img_h, img_w, device=self.device)

    def patch_features_device(self, images: torch.Tensor) -> torch.Tensor:
        """[B, H, W, 3] uint8 device tensor -> [B, 16, pw, C] fp32 device tensor (one batched forward)."""
        B, H, W, _ = images.shape
        if self.image_shape != (H, W) or (H, W) not in self._models:
            self.create_transform_(H, W)
        return self._models[(H, W)].forward(images)

    def get_image_features(self, image, upsample: bool = False, cache_file="") -> np.ndarray:
        features = None
        if cache_file:  # image_features.py:85-88
            cache_file = Path(cache_file)
            cache_file = cache_file.parent / f"{cache_file.stem}_{self.use_featup}_{upsample}.npy"
            if cache_file.exists():
                features = np.load(cache_file, allow_pickle=True)
        if features is None:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            H, W = image.shape[:2]
            img = torch.from_numpy(image).to(self.device).unsqueeze(0)
            grid = self.patch_features_device(img)[0]
            if upsample:
                vv, uu = torch.meshgrid(torch.arange(H, dtype=torch.int32, device=self.device),
                                        torch.arange(W, dtype=torch.int32, device=self.device), indexing="ij")
                k = H * W
                out = torch.zeros((k, grid.shape[-1]), dtype=torch.float32, device=self.device)
                filled = torch.zeros(k, dtype=torch.uint8, device=self.device)
                ops.gather_bilinear(grid.contiguous(), H, W, 0, None, uu.reshape(-1).contiguous(),
                                    vv.reshape(-1).contiguous(), torch.arange(k, dtype=torch.int64, device=self.device),
                                    None, out, filled)
                features = out.reshape(H, W, -1).cpu().numpy()
            else:
                features = grid.cpu().numpy()
            if cache_file:
                cache_file.parent.mkdir(parents=True, exist_ok=True)
                np.save(cache_file, features)
        return features
