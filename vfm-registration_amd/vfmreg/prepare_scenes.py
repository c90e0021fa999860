"""Mirror of prepare_scenes.create_descriptors (src/vfm-reg/src/prepare_scenes.py:50-107):
lift image features onto LiDAR points.

    create_descriptors(image_files, sequence, feature_generator, pcl[N,3]) -> [N, C] float32

Reference flow: per camera full-resolution features (image_features.py:104-110: a 2.9 GB tensor per
1600x1200 camera), zero them at black pixels (PS:57-62), project (PS:76), gather feat[v,u] per point
in a Python loop (PS:85-91), first camera wins (PS:96-101).  Here: ONE batched ViT forward for all
cameras, then per camera (in ``sequence.cameras`` priority order) a projection kernel and a fused
"interpolate at the pixel + black test + first-camera-wins scatter" kernel; the upsampled tensor is
never materialised.  ``sequence`` needs ``read_images(filenames=...) -> {camera: HxWx3 uint8}``,
``cameras`` and ``project_pcl_to_image`` (the vfmreg.dataloader classes); ``feature_generator`` is a
vfmreg.image_features.ImageFeatureGenerator, or any object with
``get_image_features(image, upsample=True) -> HxWxC`` (then the gather reads that map).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from .dataloader import NCLT


_FORCE_PER_CAMERA = False  # tests: run the per-camera path (project -> compact -> gather per camera)


def create_descriptors(image_files, sequence, feature_generator, pcl, images=None) -> np.ndarray:
    images = images if images is not None else sequence.read_images(filenames=image_files)
    cams = list(images.keys())  # the reference iterates images.items() (PS:70): dict order = camera priority
    dev = "cuda"
    n = pcl.shape[0]
    # PS:69: np.insert(pcl, 3, 1, axis=1).T (float32 xyz promoted to fp64 inside the projection)
    pcl_h = np.insert(np.asarray(pcl)[:, :3], 3, values=1, axis=1).T
    pcl_d = torch.from_numpy(np.ascontiguousarray(pcl_h, dtype=np.float64)).to(dev)
    img_d = {c: torch.from_numpy(np.ascontiguousarray(images[c], dtype=np.uint8)).to(dev) for c in cams}
    fused = hasattr(feature_generator, "patch_features_device")
    if fused:
        shapes = {tuple(images[c].shape) for c in cams}
        if len(shapes) == 1:
            grids = feature_generator.patch_features_device(torch.stack([img_d[c] for c in cams]))
            grid = {c: grids[i] for i, c in enumerate(cams)}
        else:
            grid = {c: feature_generator.patch_features_device(img_d[c].unsqueeze(0))[0] for c in cams}
    else:
        grid = {c: torch.from_numpy(np.ascontiguousarray(feature_generator.get_image_features(images[c], upsample=True),
                                                         dtype=np.float32)).to(dev) for c in cams}
    C = next(iter(grid.values())).shape[-1]
    # (vfm_lift_multicam writes every row -- zeros where nothing is seen --; the per-camera path below writes only what it claims)
    desc = torch.zeros((n, C), dtype=torch.float32, device=dev)
    filled = torch.zeros(n, dtype=torch.uint8, device=dev)
    is_nclt = isinstance(sequence, NCLT) or getattr(sequence, "rotate_images", False)
    if hasattr(sequence, "projection_params") and len(cams) <= ops.LIFT_MAX_CAMS and not _FORCE_PER_CAMERA:
        # projection fused with the gather, all cameras in one launch (same device code as the loop below)
        specs = []
        for c in cams:
            raw = img_d[c]
            proj_img = torch.rot90(raw, 1, (0, 1)).contiguous() if is_nclt else raw  # PS:73-74
            q = dict(sequence.projection_params(c, proj_img.shape))
            q.update(proj_image=proj_img if q.pop("needs_image", False) else None, grid=grid[c].contiguous(),
                     Hup=raw.shape[0], Wup=raw.shape[1], rot_mode=1 if is_nclt else 0, raw_image=raw)
            specs.append(q)
        ops.lift_multicam(pcl_d, specs, desc, filled)
        return desc.cpu().numpy()
    for c in cams:
        raw = img_d[c]
        H, W = raw.shape[0], raw.shape[1]
        if is_nclt:
            proj_img = torch.rot90(raw, 1, (0, 1)).contiguous()  # PS:73-74 cv2.ROTATE_90_COUNTERCLOCKWISE
        else:
            proj_img = raw
        try:
            u, v, idx, cnt = sequence.project_pcl_to_image(pcl_h, proj_img, c, _device_inputs=(pcl_d, proj_img))
        except TypeError:  # a duck-typed sequence with the reference's plain signature (numpy in / out)
            un, vn, idn = sequence.project_pcl_to_image(pcl_h, proj_img.cpu().numpy(), c)
            u = torch.from_numpy(np.ascontiguousarray(un, dtype=np.int32)).to(dev)
            v = torch.from_numpy(np.ascontiguousarray(vn, dtype=np.int32)).to(dev)
            idx = torch.from_numpy(np.ascontiguousarray(idn, dtype=np.int64)).to(dev)
            cnt = None
            if len(idn) == 0:
                continue
        ops.gather_bilinear(grid[c].contiguous(), H, W, 1 if is_nclt else 0, raw, u, v, idx, cnt, desc, filled)
    return desc.cpu().numpy()
