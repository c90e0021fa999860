"""Synthetic scan/map pairs of SURVEY.md section 8 D.2 (the benchmark's and the tests' inputs).

map : M points, xyz ~ U([-60,60]^2 x [-3,12]) fp32, descriptors randn(M, D) row-normalised.
scan: N points = random subset of the map moved into the scan frame by T_gt^-1 plus N(0, 0.02 m)
      noise; descriptor = matched map descriptor + sigma * randn, renormalised (inlier cosine
      ~ 0.9); a fraction rho of the rows is replaced by fresh random unit vectors (cosine ~ 0,
      rejected by the 0.8 threshold of registration_node.py:418).
T_gt: yaw ~ U(+-180 deg), roll/pitch ~ N(0, 2 deg), t_xy ~ N(0, 10 m), t_z ~ N(0, 1 m)
      (mirrors registration_node.py:847-853).  Pair p uses seed 42 + p.
"""
from __future__ import annotations

import math

import numpy as np


def random_pose(rng: np.random.Generator) -> np.ndarray:
    yaw = rng.uniform(-math.pi, math.pi)
    roll, pitch = np.deg2rad(rng.normal(0.0, 2.0, 2))
    cr, sr, cp, sp, cy, sy = math.cos(roll), math.sin(roll), math.cos(pitch), math.sin(pitch), math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [rng.normal(0, 10), rng.normal(0, 10), rng.normal(0, 1)]
    return T


def make_pair(n: int, m: int, d: int = 384, seed: int = 42, outlier: float = 0.5, inlier_cos: float = 0.9,
              noise_m: float = 0.02):
    """numpy generator (CPU). Returns dict(q_desc [n,d] f32, q_xyz [n,3] f64, b_desc [m,d] f32,
    b_xyz [m,3] f64, T_gt [4,4], match [n] int64 (-1 for outlier rows))."""
    rng = np.random.default_rng(seed)
    b_xyz = np.c_[rng.uniform(-60, 60, m), rng.uniform(-60, 60, m), rng.uniform(-3, 12, m)].astype(np.float32)
    b_desc = rng.standard_normal((m, d), dtype=np.float32)
    b_desc /= np.linalg.norm(b_desc, axis=1, keepdims=True)
    T = random_pose(rng)
    pick = rng.choice(m, size=n, replace=(n > m))
    R, t = T[:3, :3], T[:3, 3]
    q_xyz = (b_xyz[pick].astype(np.float64) - t) @ R  # R^T (p - t)
    q_xyz = q_xyz + rng.normal(0, noise_m, q_xyz.shape)
    sigma = math.sqrt((1.0 / inlier_cos ** 2 - 1.0) / d)
    q_desc = b_desc[pick] + sigma * rng.standard_normal((n, d), dtype=np.float32)
    is_out = rng.random(n) < outlier
    q_desc[is_out] = rng.standard_normal((int(is_out.sum()), d), dtype=np.float32)
    q_desc /= np.linalg.norm(q_desc, axis=1, keepdims=True)
    match = np.where(is_out, -1, pick).astype(np.int64)
    return dict(q_desc=np.ascontiguousarray(q_desc, dtype=np.float32), q_xyz=np.ascontiguousarray(q_xyz),
                b_desc=np.ascontiguousarray(b_desc, dtype=np.float32),
                b_xyz=np.ascontiguousarray(b_xyz.astype(np.float64)), T_gt=T, match=match)


def make_pair_device(n: int, m: int, d: int = 384, seed: int = 42, device="cuda", outlier: float = 0.5,
                     inlier_cos: float = 0.9, noise_m: float = 0.02):
    """Same distribution generated on the device with torch (the 20k x 200k x 384 benchmark pair is
    230 M floats; the host is not the bottleneck).  Returns torch tensors + T_gt (numpy)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rng = np.random.default_rng(seed)
    T = random_pose(rng)
    b_xyz = torch.rand((m, 3), generator=g, device=device, dtype=torch.float32)
    b_xyz = b_xyz * torch.tensor([120.0, 120.0, 15.0], device=device) + torch.tensor([-60.0, -60.0, -3.0], device=device)
    b_desc = torch.randn((m, d), generator=g, device=device, dtype=torch.float32)
    b_desc /= b_desc.norm(dim=1, keepdim=True)
    pick = torch.randint(0, m, (n,), generator=g, device=device) if n > m else torch.randperm(m, generator=g, device=device)[:n]
    Tt = torch.tensor(T, device=device, dtype=torch.float64)
    q_xyz = (b_xyz[pick].double() - Tt[:3, 3]) @ Tt[:3, :3]
    q_xyz = q_xyz + noise_m * torch.randn(q_xyz.shape, generator=g, device=device, dtype=torch.float64)
    sigma = math.sqrt((1.0 / inlier_cos ** 2 - 1.0) / d)
    q_desc = b_desc[pick] + sigma * torch.randn((n, d), generator=g, device=device, dtype=torch.float32)
    is_out = torch.rand(n, generator=g, device=device) < outlier
    fresh = torch.randn((n, d), generator=g, device=device, dtype=torch.float32)
    q_desc = torch.where(is_out[:, None], fresh, q_desc)
    q_desc /= q_desc.norm(dim=1, keepdim=True)
    match = torch.where(is_out, torch.full_like(pick, -1), pick)
    return dict(q_desc=q_desc.contiguous(), q_xyz=q_xyz.contiguous(), b_desc=b_desc.contiguous(),
                b_xyz=b_xyz.double().contiguous(), T_gt=T, match=match)


def lifted_map(m: int, d: int, clouds: int, cams: int, gh: int, gw: int, seed: int, view_noise: float, device="cuda",
               revisit: int = 0):
    """Map descriptors that look like LIFTED ones (prepare_scenes.py:50-107 + image_features.py:104-108): row r belongs to
    (cloud, camera) image k and is the bilinear sample of that image's gh x gw patch grid at a random position.  Images of
    different clouds that look at the same place share a smooth scene field: grid(k) = scene_grid(camera) + view_noise *
    randn.  ``revisit`` > 0: the same physical points are seen again by every cloud (that many pixel positions per camera)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    scene = torch.randn((cams, gh, gw, d), generator=g, device=device)
    K = clouds * cams
    img = torch.randint(0, K, (m,), generator=g, device=device)
    cam = img % cams
    noise_seed = torch.randn((K, gh, gw, d), generator=g, device=device) * view_noise
    if revisit:
        py = torch.rand((cams, revisit), generator=g, device=device) * (gh - 1 - 1e-3)
        px = torch.rand((cams, revisit), generator=g, device=device) * (gw - 1 - 1e-3)
        which = torch.randint(0, revisit, (m,), generator=g, device=device)
        y, x = py[cam, which], px[cam, which]
    else:
        y = torch.rand(m, generator=g, device=device) * (gh - 1 - 1e-3)
        x = torch.rand(m, generator=g, device=device) * (gw - 1 - 1e-3)
    i, j = y.long(), x.long()
    fy, fx = (y - i)[:, None], (x - j)[:, None]

    def at(ii, jj):
        return scene[cam, ii, jj] + noise_seed[img, ii, jj]
    out = at(i, j) * (1 - fy) * (1 - fx) + at(i + 1, j) * fy * (1 - fx) + at(i, j + 1) * (1 - fy) * fx + at(i + 1, j + 1) * fy * fx
    return out.float().contiguous()


def make_lifted_pair_device(n: int, m: int, d: int = 384, seed: int = 42, device="cuda", clouds: int = 10,
                            view_noise: float = 0.1, common: float = 0.0, revisit: int = 0):
    """A D.2 pair (same geometry, same planted matches, same outlier rows) whose MAP descriptors are lifted ones
    (``lifted_map``) and whose scan descriptors are the matched map rows + 0.3 rms noise.  ``common`` > 0 adds one shared
    vector of that many rms to every row (scan and map): descriptors that are all alike, as a ViT's patch tokens are."""
    import torch
    base = make_pair_device(n, m, d, seed=seed, device=device)
    b = lifted_map(m, d, clouds, 6, 16, 21, seed + 7, view_noise, device, revisit)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    rms = b.pow(2).mean().sqrt()
    pick = base["match"].clamp(min=0)
    q = b[pick] + 0.3 * rms * torch.randn((n, d), generator=g, device=device)
    q = torch.where((base["match"] < 0)[:, None], rms * torch.randn((n, d), generator=g, device=device), q)
    if common > 0:
        mu = common * rms * torch.randn((1, d), generator=g, device=device)
        b, q = b + mu, q + mu
    base["b_desc"], base["q_desc"] = b.contiguous(), q.contiguous()
    return base
