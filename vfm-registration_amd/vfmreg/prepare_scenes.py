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


def create_descriptors(image_files, sequence, feature_generator, pcl, images=None, _grids=None, _img_d=None) -> np.ndarray:
    """``_grids`` (create_descriptors_batch): the cameras' patch features, already computed in a larger batch -- {camera: device
    tensor [16, pw, C]}; ``_img_d``: the cameras' images where that batch already put them on the device ({camera: uint8 tensor})."""
    images = images if images is not None else sequence.read_images(filenames=image_files)
    cams = list(images.keys())  # the reference iterates images.items() (PS:70): dict order = camera priority
    dev = "cuda"
    n = pcl.shape[0]
    # PS:69: np.insert(pcl, 3, 1, axis=1).T (float32 xyz promoted to fp64 inside the projection)
    pcl_h = np.insert(np.asarray(pcl)[:, :3], 3, values=1, axis=1).T
    pcl_d = torch.from_numpy(np.ascontiguousarray(pcl_h, dtype=np.float64)).to(dev)
    img_d = _img_d if _img_d is not None else {c: torch.from_numpy(np.ascontiguousarray(images[c], dtype=np.uint8)).to(dev) for c in cams}
    fused = hasattr(feature_generator, "patch_features_device")
    if _grids is not None:
        grid = _grids
    elif fused:
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


def create_descriptors_batch(image_files_list, sequence, feature_generator, pcls, images_list=None, clouds_per_forward: int = 28):
    """``create_descriptors`` for a LIST of clouds of one sequence -- what prepare_scenes.main's loops over ~170 map clouds and the
    scans of a scene call one cloud at a time (PS:129-163).  The ViT forward is batched over ``clouds_per_forward`` clouds x their
    cameras (6 x 28 = 168 images of 1200 x 1600 per call: 0.171 ms per cloud against 0.52 for one cloud per call; the wide GEMMs move to
    the LDS-tiled kernels, QKV + attention to one workgroup per (image, head) and, from two full rounds of workgroups on, fc1 -> GELU -> fc2 to
    one workgroup per 128 tokens, csrc/vit.hip -- 168 images are 1008 + 462 such workgroups on 256 compute units: 28.5 us per image; 14 clouds
    (84 images, two rounds of the first kind, one of the second): 29.8; round 5's 15 clouds: 34.7;
    profiles/r06_ab_vit_fused_qkv_sweep.txt, profiles/r06_ab_vit_fused_mlp_sweep.txt);
    projection and lifting stay per cloud.  A larger batch does not change a single bit of any
    cloud's features (tests/test_gpu_vit.py, tools/time_vit_batch.py), so every returned array equals ``create_descriptors`` of
    that cloud.  Falls back to the per-cloud call for generators without ``patch_features_device`` or cameras of mixed sizes."""
    n_clouds = len(pcls)
    # images are read -- and uploaded -- one group of clouds at a time (~170 clouds x 6 cameras x 5.8 MB read up front were
    # ~6 GB of host memory for one scene, and every image went to the device twice: once for the ViT batch, once for the lifting)
    def images_of(i):
        return images_list[i] if images_list is not None else sequence.read_images(filenames=image_files_list[i])
    out = [None] * n_clouds
    batched = hasattr(feature_generator, "patch_features_device") and clouds_per_forward > 1
    for i0 in range(0, n_clouds, max(clouds_per_forward, 1)):
        group = range(i0, min(n_clouds, i0 + max(clouds_per_forward, 1)))
        imgs = {i: images_of(i) for i in group}
        shapes = {tuple(np.asarray(im).shape) for i in group for im in imgs[i].values()}
        if not batched or len(shapes) != 1:
            for i in group:
                out[i] = create_descriptors(None, sequence, feature_generator, pcls[i], images=imgs[i])
            continue
        keys = [(i, c) for i in group for c in imgs[i].keys()]
        batch = torch.stack([torch.from_numpy(np.ascontiguousarray(imgs[i][c], dtype=np.uint8)) for i, c in keys]).to("cuda")
        grids = feature_generator.patch_features_device(batch)
        per_cloud = {i: ({}, {}) for i in group}
        for k, (i, c) in enumerate(keys):
            per_cloud[i][0][c] = grids[k]
            per_cloud[i][1][c] = batch[k]
        for i in group:
            out[i] = create_descriptors(None, sequence, feature_generator, pcls[i], images=imgs[i], _grids=per_cloud[i][0],
                                        _img_d=per_cloud[i][1])
    return out


def prepare_scene(dataset_dir, scene_data: dict, Dataset, feature_generator, date_idx: int, output_filename=None,
                  clouds_per_forward: int = 28, voxel_down_sample=None):
    """prepare_scenes.main without its argument parsing and progress bars (PS:110-171): the map clouds of a scene (voxelised at
    0.2 m, PS:134) and its scans (0.1 m, PS:152) with their lifted descriptors, through ``create_descriptors_batch``; written as the
    reference's HDF5 scene file when ``output_filename`` is given (PS:165-166 -> evaluation.save_scene).  ``Dataset`` is one of
    the vfmreg.dataloader classes (or anything with ``read_pcl`` / ``read_images`` / projection): reading the dataset's files is
    the sequence object's business, as in the reference.  Returns (sequences, map_poses, map_clouds, seq_poses, seq_clouds)."""
    from pathlib import Path
    from .voxelization import voxel_down_sample as _vds
    vds = voxel_down_sample or _vds
    dataset_dir = Path(dataset_dir)
    sequences = [scene_data["mapping"]["point_clouds"][date_idx].split("/")[1]]                      # PS:122-125
    for seq in scene_data["registration"]:
        sequences.append(seq["point_cloud"].split("/")[date_idx])
    map_sequence = Dataset(sequences[0], dataset_dir, high_level_api=True)                           # PS:128
    pcls = []
    for pcl_file in scene_data["mapping"]["point_clouds"]:                                           # PS:131-134
        pcl = map_sequence.read_pcl(filename=dataset_dir / pcl_file)
        pcls.append(vds(pcl, 0.2).astype(pcl.dtype))
    files = [[dataset_dir / f for f in fs] for fs in scene_data["mapping"]["images"]]                # PS:136
    descs = create_descriptors_batch(files, map_sequence, feature_generator, pcls, clouds_per_forward=clouds_per_forward)
    map_point_clouds = [np.c_[p, d] for p, d in zip(pcls, descs)]                                    # PS:138
    map_poses = [np.array(pose) for pose in scene_data["mapping"]["poses"]]                          # PS:143
    seq_point_clouds, seq_poses = [], []
    for i, registration in enumerate(scene_data["registration"]):                                    # PS:148-163 (a sequence per scan)
        registration_sequence = Dataset(sequences[i + 1], dataset_dir, high_level_api=True)
        pcl = registration_sequence.read_pcl(filename=dataset_dir / registration["point_cloud"])
        pcl = vds(pcl, 0.1).astype(pcl.dtype)
        image_files = [dataset_dir / f for f in registration["images"]]
        seq_point_clouds.append(np.c_[pcl, create_descriptors(image_files, registration_sequence, feature_generator, pcl)])
        seq_poses.append(np.array(registration["pose"]))
    if output_filename is not None:
        from .evaluation import save_scene
        save_scene(output_filename, sequences, map_poses, map_point_clouds, seq_poses, seq_point_clouds)
    return sequences, map_poses, map_point_clouds, seq_poses, seq_point_clouds
