"""CPU oracle for the VFM-Registration hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``vfm-registration_amd/`` does.  It wraps ``oracle/libvfm_oracle.so``
(the C restatement, see ``vfm_oracle.c`` for per-function reference citations and the
pinned / unpinned parity status) and restates the Python-level reference functions in numpy.

Citations use SURVEY.md's shorthand: RN = src/vfm-reg/src/registration_node.py,
PS = src/vfm-reg/src/prepare_scenes.py, IF = src/vfm-reg/src/vfm_reg/image_features.py,
UT = src/vfm-reg/src/vfm_reg/utils.py, VHM = src/kiss-icp/cpp/kiss_icp/core/VoxelHashMap.cpp.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB: Optional[C.CDLL] = None

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


def build() -> Path:
    """Compile the C restatement with gcc (oracle/Makefile)."""
    subprocess.run(["make", "-s", "-C", str(_HERE)], check=True)
    return _HERE / "libvfm_oracle.so"


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        so = _HERE / "libvfm_oracle.so"
        src = _HERE / "vfm_oracle.c"
        if not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
            build()
        _LIB = C.CDLL(str(so))
        _LIB.orc_threshold_compact.restype = C.c_int64
        _LIB.orc_project.restype = C.c_int64
        _LIB.orc_voxel_first.restype = C.c_int64
        _LIB.orc_kabsch.restype = C.c_int
        _LIB.orc_num_threads.restype = C.c_int
    return _LIB


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def num_threads() -> int:
    return int(lib().orc_num_threads())


# ----------------------------------------------------------------------------- A5 matching
def l2norm_rows(x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """faiss::fvec_renorm_L2 per row (VHM:474,480). Returns (normalised fp32, inv_norm fp32)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    out = np.empty_like(x)
    inv = np.empty(n, dtype=np.float32)
    lib().orc_l2norm_rows_f32(_p(x, _f32p), C.c_int64(n), C.c_int(d), _p(out, _f32p), _p(inv, _f32p))
    return out, inv


def match_ip_top1_bruteforce(qn: np.ndarray, bn: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """IndexFlatIP.search(k=1) (VHM:486-495) decided in fp64 over ALL pairs (small sizes only)."""
    qn = np.ascontiguousarray(qn, dtype=np.float32)
    bn = np.ascontiguousarray(bn, dtype=np.float32)
    n, d = qn.shape
    m = bn.shape[0]
    idx = np.empty(n, dtype=np.int64)
    sim = np.empty(n, dtype=np.float32)
    lib().orc_match_ip_top1(_p(qn, _f32p), C.c_int64(n), _p(bn, _f32p), C.c_int64(m), C.c_int(d),
                            _p(idx, _i64p), _p(sim, _f32p))
    return idx, sim


def match_ip_top1(qn: np.ndarray, bn: np.ndarray, block: int = 1024,
                  window: Optional[float] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Same decision as ``match_ip_top1_bruteforce`` at sizes where all-pairs fp64 is too slow.

    A BLAS fp32 ``Q @ B.T`` proposes, per row, every map row whose fp32 score is within
    ``window`` of the row maximum; ``orc_match_ip_candidates`` then decides exactly among them.
    The fp32 GEMM error is bounded by gamma_d * sum|q_k b_k| <= d * 2^-23 * |q||b| (Higham,
    Accuracy and Stability, eq. 3.5; 1.05 covers |x| = 1 + O(1e-6)), so with
    window >= 2 * bound the true fp64 arg-max is always among the candidates.
    """
    qn = np.ascontiguousarray(qn, dtype=np.float32)
    bn = np.ascontiguousarray(bn, dtype=np.float32)
    n, d = qn.shape
    m = bn.shape[0]
    if window is None:
        qmax = float(np.sqrt((qn.astype(np.float64) ** 2).sum(1).max())) if n else 1.0
        bmax = float(np.sqrt((bn.astype(np.float64) ** 2).sum(1).max())) if m else 1.0
        window = 2.0 * 1.05 * d * 2.0 ** -23 * max(qmax * bmax, 1e-30)
    ptr = np.zeros(n + 1, dtype=np.int64)
    cands = []
    for s in range(0, n, block):
        sc = qn[s:s + block] @ bn.T
        mx = sc.max(axis=1, keepdims=True)
        rows, cols = np.nonzero(sc >= mx - np.float32(window))
        cnt = np.bincount(rows, minlength=sc.shape[0])
        ptr[s + 1:s + 1 + sc.shape[0]] = cnt
        cands.append(cols.astype(np.int64))
    ptr = np.cumsum(ptr)
    cand = np.ascontiguousarray(np.concatenate(cands) if cands else np.zeros(0, np.int64))
    idx = np.empty(n, dtype=np.int64)
    sim = np.empty(n, dtype=np.float32)
    lib().orc_match_ip_candidates(_p(qn, _f32p), C.c_int64(n), _p(bn, _f32p), C.c_int(d),
                                  _p(ptr, _i64p), _p(cand, _i64p), _p(idx, _i64p), _p(sim, _f32p))
    return idx, sim


def threshold_compact(sim: np.ndarray, thr: float) -> np.ndarray:
    """valid = !(D < thr) in query order (VHM:501-511, 587-600)."""
    sim = np.ascontiguousarray(sim, dtype=np.float32)
    keep = np.empty(sim.shape[0], dtype=np.int64)
    k = lib().orc_threshold_compact(_p(sim, _f32p), C.c_int64(sim.shape[0]), C.c_double(thr),
                                    _p(keep, _i64p))
    return keep[:k].copy()


def get_vfm_correspondences(query: np.ndarray, map_pts: np.ndarray, thr: float,
                            bruteforce: bool = False):
    """VoxelHashMap::GetVFMCorrespondences (VHM:461-626) on explicit arrays.

    query N x (3+D), map_pts M x (3+D) (the PointcloudN() dump).  Returns
    (src_xyz, tgt_xyz, query_idx, map_idx, sim) -- the reference returns only the first two.
    """
    qn, _ = l2norm_rows(query[:, 3:].astype(np.float32))
    bn, _ = l2norm_rows(map_pts[:, 3:].astype(np.float32))
    idx, sim = (match_ip_top1_bruteforce if bruteforce else match_ip_top1)(qn, bn)
    keep = threshold_compact(sim, thr)
    return (query[keep, :3].astype(np.float64), map_pts[idx[keep], :3].astype(np.float64), keep,
            idx[keep], sim)


# ----------------------------------------------------------------------------- A6 mutual NN
def nn_l2(a: np.ndarray, b: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    idx = np.empty(a.shape[0], dtype=np.int64)
    dist = np.empty(a.shape[0], dtype=np.float64)
    lib().orc_nn_l2(_p(a, _f32p), C.c_int64(a.shape[0]), _p(b, _f32p), C.c_int64(b.shape[0]),
                    C.c_int(a.shape[1]), _p(idx, _i64p), _p(dist, _f64p))
    return idx, dist


def find_correspondences(feats0: np.ndarray, feats1: np.ndarray, n_points: int = 5000,
                         mutual_filter: bool = True):
    """RN:482-538 (nested find_correspondences), cKDTree replaced by exact brute force."""
    nns01, dists = nn_l2(feats0, feats1)
    idx0 = np.arange(len(nns01))
    idx1 = nns01
    if not mutual_filter:
        n = min(n_points, len(dists) - 1)
        top = np.argpartition(dists, n)[:n]
        return idx0[top], idx1[top]
    nns10, _ = nn_l2(feats1, feats0)
    mutual = nns10[idx1] == idx0
    return idx0[mutual], idx1[mutual]


# ----------------------------------------------------------------------------- A8/A9 RANSAC
def kabsch(A: np.ndarray, B: np.ndarray, w: Optional[np.ndarray] = None,
           denom_eps: float = 0.0) -> Tuple[np.ndarray, bool]:
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    T = np.empty((4, 4), dtype=np.float64)
    wp = None
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.float64)
        wp = _p(w, _f64p)
    ok = lib().orc_kabsch(_p(A, _f64p), _p(B, _f64p), wp, C.c_int64(A.shape[0]),
                          C.c_double(denom_eps), _p(T, _f64p))
    return T, bool(ok)


def kabsch_svd(A: np.ndarray, B: np.ndarray, w: Optional[np.ndarray] = None) -> np.ndarray:
    """Textbook Kabsch/Umeyama (no scaling) via numpy SVD: independent check of ``kabsch``."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    w = np.ones(len(A)) if w is None else np.asarray(w, np.float64)
    ma = (w[:, None] * A).sum(0) / w.sum()
    mb = (w[:, None] * B).sum(0) / w.sum()
    S = ((B - mb) * w[:, None]).T @ (A - ma) / w.sum()
    U, _, Vt = np.linalg.svd(S)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U) * np.linalg.det(Vt))])
    R = U @ D @ Vt
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mb - R @ ma
    return T


def philox(ctr: int, seed: int) -> np.ndarray:
    out = (C.c_uint32 * 4)()
    lib().orc_philox(C.c_uint32(ctr), C.c_uint64(seed), out)
    return np.array(list(out), dtype=np.uint32)


class RansacResult:
    """Duck-type of open3d.pipelines.registration.RegistrationResult (RN:328)."""

    def __init__(self, T, fitness, rmse, mask, best_hyp, hyp_fit=None, hyp_rmse=None):
        self.transformation = T
        self.fitness = fitness
        self.inlier_rmse = rmse
        self.inlier_mask = mask
        self.best_hyp = best_hyp
        self.hyp_fit = hyp_fit
        self.hyp_rmse = hyp_rmse


def ransac_corr(src: np.ndarray, tgt: np.ndarray, corres: np.ndarray, max_dist: float, n_iter: int,
                seed: int = 42, per_hyp: bool = False) -> RansacResult:
    """registration_ransac_based_on_correspondence as called at RN:319-327."""
    src = np.ascontiguousarray(src, dtype=np.float64)
    tgt = np.ascontiguousarray(tgt, dtype=np.float64)
    corres = np.ascontiguousarray(corres, dtype=np.int32)
    Cn = corres.shape[0]
    T = np.empty((4, 4), dtype=np.float64)
    fit = C.c_double(0)
    rmse = C.c_double(0)
    best = C.c_int32(-1)
    mask = np.zeros(max(Cn, 1), dtype=np.uint8)
    hf = np.empty(n_iter, dtype=np.float64) if per_hyp else None
    hr = np.empty(n_iter, dtype=np.float64) if per_hyp else None
    lib().orc_ransac_corr(_p(src, _f64p), _p(tgt, _f64p), _p(corres, _i32p), C.c_int64(Cn),
                          C.c_double(max_dist), C.c_int32(n_iter), C.c_uint64(seed), _p(T, _f64p),
                          C.byref(fit), C.byref(rmse), _p(mask, _u8p), C.byref(best),
                          _p(hf, _f64p) if per_hyp else None, _p(hr, _f64p) if per_hyp else None)
    return RansacResult(T, fit.value, rmse.value, mask[:Cn].copy(), best.value, hf, hr)


def orthogonalize_rotation(T: np.ndarray) -> np.ndarray:
    """RN:331-336: Newton iteration R <- 1.5 R - 0.5 R R^T R until |1 - det R| <= 1e-12."""
    T = np.array(T, dtype=np.float64, copy=True)
    R = T[:3, :3]
    while np.abs(1 - np.linalg.det(R)) > 1e-12:
        R = 3 / 2 * R - 1 / 2 * R @ R.T @ R
    T[:3, :3] = R
    return T


def compute_errors(pose: np.ndarray, gt: np.ndarray) -> Tuple[float, float]:
    """RN:997-1019: (RTE [m], RRE [deg])."""
    rte = float(np.linalg.norm(pose[:3, 3] - gt[:3, 3]))
    c = (np.trace(pose[:3, :3].T @ gt[:3, :3]) - 1) / 2
    rre = float(np.abs(np.arccos(np.clip(c, -1, 1))) * 180 / np.pi)
    return rte, rre


# ----------------------------------------------------------------------------- A2 projection
def project(mode: int, pcl4xn: np.ndarray, mats, fc, subsample: float, win, image: Optional[np.ndarray],
            H: int, W: int):
    """Dataset.project_pcl_to_image: mode 0 NCLT:311-366, 1 OXF:330-363, 2 KIT:110-125."""
    pcl = np.ascontiguousarray(pcl4xn, dtype=np.float64)
    n = pcl.shape[1]
    M = [np.ascontiguousarray(np.asarray(m, dtype=np.float64).ravel()) if m is not None
         else np.zeros(16) for m in (list(mats) + [None, None, None])[:3]]
    M = [np.concatenate([m, np.zeros(16 - m.size)]) if m.size < 16 else m for m in M]
    fcv = np.ascontiguousarray(fc if fc is not None else np.zeros(4), dtype=np.float64)
    winv = np.ascontiguousarray(win if win is not None else np.zeros(4), dtype=np.int64)
    u = np.empty(n, dtype=np.int64)
    v = np.empty(n, dtype=np.int64)
    idx = np.empty(n, dtype=np.int64)
    img = None
    if image is not None:
        image = np.ascontiguousarray(image, dtype=np.uint8)
        img = _p(image, _u8p)
    k = lib().orc_project(C.c_int(mode), _p(pcl, _f64p), C.c_int64(n), _p(M[0], _f64p),
                          _p(M[1], _f64p), _p(M[2], _f64p), _p(fcv, _f64p), C.c_double(subsample),
                          _p(winv, _i64p), img, C.c_int64(H), C.c_int64(W), _p(u, _i64p),
                          _p(v, _i64p), _p(idx, _i64p))
    return u[:k].copy(), v[:k].copy(), idx[:k].copy()


# ----------------------------------------------------------------------------- A3 lifting
def gather_bilinear(grid: np.ndarray, Hup: int, Wup: int, rot_mode: int, u: np.ndarray,
                    v: np.ndarray) -> np.ndarray:
    """F.interpolate(bilinear, align_corners=False) to Hup x Wup (IF:104-108) then [v,u] (PS:85)."""
    grid = np.ascontiguousarray(grid, dtype=np.float32)
    gh, gw, Cc = grid.shape
    u = np.ascontiguousarray(u, dtype=np.int64)
    v = np.ascontiguousarray(v, dtype=np.int64)
    out = np.empty((len(u), Cc), dtype=np.float32)
    lib().orc_gather_bilinear(_p(grid, _f32p), C.c_int64(gh), C.c_int64(gw), C.c_int64(Cc),
                              C.c_int64(Hup), C.c_int64(Wup), C.c_int(rot_mode), _p(u, _i64p),
                              _p(v, _i64p), C.c_int64(len(u)), _p(out, _f32p))
    return out


def create_descriptors(n_points: int, cams: list) -> np.ndarray:
    """create_descriptors (PS:50-107) given, per camera in priority order, a dict with
    ``grid`` (gh x gw x C patch features), ``Hup, Wup`` (size of the un-rotated upsampled map),
    ``rot_mode``, ``black`` (H x W bool, pixels whose RGB is all zero, in the un-rotated image;
    PS:57-62) and the projection result ``u, v, idx``.  First camera wins (PS:96-101)."""
    Cc = cams[0]["grid"].shape[2]
    desc = np.zeros((n_points, Cc), dtype=np.float32)
    filled = np.zeros(n_points, dtype=bool)
    for cam in cams:
        u, v, idx = cam["u"], cam["v"], cam["idx"]
        if len(idx) == 0:
            continue
        f = gather_bilinear(cam["grid"], cam["Hup"], cam["Wup"], cam["rot_mode"], u, v)
        if cam.get("black") is not None:
            if cam["rot_mode"] == 1:
                blk = cam["black"][u, cam["Wup"] - 1 - v]
            else:
                blk = cam["black"][v, u]
            f[blk] = 0.0
        new = ~filled[idx]
        desc[idx[new]] = f[new]
        filled[idx[new]] = True
    return desc


def transform_pcl(pcl: np.ndarray, T: np.ndarray) -> np.ndarray:
    """UT:47-54."""
    xyz = np.ascontiguousarray(pcl[:, :3], dtype=np.float64)
    out = np.empty_like(xyz)
    Tm = np.ascontiguousarray(T, dtype=np.float64)
    lib().orc_transform_xyz(_p(xyz, _f64p), C.c_int64(len(xyz)), _p(Tm, _f64p), _p(out, _f64p))
    return np.c_[out, pcl[:, 3:]].astype(pcl.dtype)


# ----------------------------------------------------------------------------- F1 voxels
def voxel_first(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1) -> np.ndarray:
    """kiss_icp VoxelDownsample (Preprocessing.cpp:50-137) / VoxelHashMap::AddPoints
    (VHM:746-757, cap max_points_per_voxel): indices of survivors in input order."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    keep = np.empty(len(pts), dtype=np.int64)
    k = lib().orc_voxel_first(_p(pts, _f64p), C.c_int64(len(pts)), C.c_int64(pts.shape[1]),
                              C.c_double(voxel_size), C.c_int64(max_per_voxel), _p(keep, _i64p))
    return keep[:k].copy()


HASH_MUL_DOWNSAMPLE = 19349663  # Preprocessing.cpp:44
HASH_MUL_MAP = 19349669         # VoxelHashMap.hpp:75


def voxel_robin(points: np.ndarray, voxel_size: float, max_per_voxel: int = 1, reserve: bool = True,
                hash_mul: int = HASH_MUL_DOWNSAMPLE, return_info: bool = False):
    """Point indices in the order the reference emits them: tsl::robin_map iteration order
    (restated operation by operation in vfm_oracle.c, v1.2.1; parity unpinned -- the header is not in
    the tree).  ``reserve=True, max_per_voxel=1, HASH_MUL_DOWNSAMPLE`` = VoxelDownsample
    (Preprocessing.cpp:50-69); ``reserve=False, max_per_voxel=K, HASH_MUL_MAP`` = a fresh
    VoxelHashMap after AddPoints, as Pointcloud()/PointcloudN() iterate it (VHM:640-676, 733-770)."""
    pts = np.ascontiguousarray(points, dtype=np.float64)
    out = np.empty(max(len(pts), 1), dtype=np.int64)
    info = np.zeros(4, dtype=np.int64)
    f = lib().orc_voxel_robin
    f.restype = C.c_int64
    k = f(_p(pts, _f64p), C.c_int64(len(pts)), C.c_int64(pts.shape[1] if pts.ndim == 2 else 3),
          C.c_double(voxel_size), C.c_int64(max_per_voxel), C.c_uint32(hash_mul),
          C.c_int64(len(pts) if reserve else -1), _p(out, _i64p), _p(info, _i64p))
    return (out[:k].copy(), info) if return_info else out[:k].copy()


def voxel_down_sample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """kiss_icp.voxelization.voxel_down_sample as the reference returns it (rows in robin_map order)."""
    points = np.asarray(points)
    return np.asarray(points[voxel_robin(points[:, :3], voxel_size)], dtype=np.float64)


def voxel_hash_map_points(points: np.ndarray, voxel_size: float, max_per_voxel: int) -> np.ndarray:
    """get_voxel_hash_map(); add_points(points); point_cloud*(): indices of the rows, in map order."""
    return voxel_robin(np.asarray(points)[:, :3], voxel_size, max_per_voxel, reserve=False, hash_mul=HASH_MUL_MAP)


def voxel_hash(vox: np.ndarray, hash_mul: int) -> np.ndarray:
    v = vox.astype(np.int32).view(np.uint32).astype(np.uint64)
    h = (v[:, 0] * 73856093) ^ (v[:, 1] * hash_mul) ^ (v[:, 2] * 83492791)
    return (h & 0xFFFFF).astype(np.int64)


def robin_order_by_clusters(hashes: np.ndarray, reserve_n: Optional[int]) -> np.ndarray:
    """Second derivation of the robin_map iteration order of distinct keys inserted in sequence (argument:
    their 20-bit hashes), structured like csrc/voxel.hip so that the decomposition the GPU uses is checked
    on the CPU against the bucket-level container simulation of vfm_oracle.c:
      * a robin-hood table with linear probing holds, cyclically, its keys sorted by home bucket, so the
        occupied bucket runs ("clusters") follow from a stable sort by ``hash & mask`` and a running maximum
        (bucket of sorted entry i = i + max_{j<=i}(home_j - j));
      * clusters never interact: replaying the insertions of one cluster's keys (arrival order, tsl's swap
        rule) inside its own window gives the container's layout there;
      * arrival order at a rehash is the old table's iteration order; a growing map is one such step per
        table generation; entries wrapping past the last bucket: redo in coordinates rotated to an empty bucket."""
    n = len(hashes)
    hashes = np.asarray(hashes, dtype=np.int64)
    order = np.zeros(0, dtype=np.int64)
    if reserve_n is not None:
        c = int(np.ceil(np.float32(reserve_n) / np.float32(0.5)))
        B = 0 if c == 0 else 1 << max(0, int(c - 1).bit_length())
    else:
        B = 0
    s = 0
    while s < n:
        thr = int(np.float32(B) * np.float32(0.5))
        if s >= thr:
            B = 2 * B if B else 2
            thr = int(np.float32(B) * np.float32(0.5))
        m = min(n, thr)
        seq = np.concatenate([order, np.arange(s, m)])
        z = 0
        for attempt in range(2):
            home = (hashes[seq] - z) & (B - 1)
            perm = np.argsort(home, kind="stable")
            hs = home[perm]
            d = hs - np.arange(m)
            cm = np.maximum.accumulate(d)
            w = int(np.count_nonzero(np.arange(m) + cm >= B))
            if w > 0:
                assert attempt == 0
                a = int(np.argmax(cm >= w + 1))
                z = a + int(cm[a]) - 1
                continue
            start = np.r_[True, d[1:] > cm[:-1]]
            cidx = np.cumsum(start) - 1
            cl_of_pos = np.empty(m, dtype=np.int64)
            cl_of_pos[perm] = cidx
            arr = seq[np.argsort(cl_of_pos, kind="stable")]           # per cluster, in arrival order
            starts = np.r_[np.nonzero(start)[0], m]
            tab = np.empty(m, dtype=np.int64)
            for c in range(len(starts) - 1):
                i0, i1 = int(starts[c]), int(starts[c + 1])
                L, base = i1 - i0, int(hs[i0])
                D = [-1] * L
                I = [0] * L
                for v in arr[i0:i1]:
                    v = int(v)
                    ib, dist = ((int(hashes[v]) - z) & (B - 1)) - base, 0
                    while True:
                        if dist > D[ib]:
                            if D[ib] < 0:
                                D[ib], I[ib] = dist, v
                                break
                            D[ib], dist = dist, D[ib]
                            I[ib], v = v, I[ib]
                        dist += 1
                        ib += 1
                tab[i0:i1] = I
            r = int(np.argmax(np.arange(m) + cm >= B - z)) if (z > 0 and np.any(np.arange(m) + cm >= B - z)) else 0
            order = np.concatenate([tab[r:], tab[:r]])
            break
        s = m
    return order


# ----------------------------------------------------------------------------- A1 ViT (torch fp32)
def vit_reference(weights: dict, img_u8: np.ndarray, patch_h: int = 16):
    """ImageFeatureGenerator.get_image_features(upsample=False) for 'dinov2', use_featup=False
    (IF:67-117): resize (bilinear, antialias off) so that the image is patch_h patches high,
    ImageNet-normalise, DINOv2 ViT-S/14 forward_features()['x_norm_patchtokens'], FeatUp
    ChannelNorm.  Plain PyTorch fp32 on CPU; floating point => tolerance parity.
    img_u8: B x H x W x 3.  Returns B x patch_h x pw x C fp32 (channels last).
    """
    import torch
    import torch.nn.functional as F

    w = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in weights.items()}
    B, H, W, _ = img_u8.shape
    ps = 14
    scale = (ps * patch_h) / H
    pw = int(scale * W / ps)
    x = torch.from_numpy(img_u8).permute(0, 3, 1, 2).float() / 255.0
    x = F.interpolate(x, size=(ps * patch_h, ps * pw), mode="bilinear", align_corners=False,
                      antialias=False)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = (x - mean) / std
    Dm = w["patch_embed.proj.weight"].shape[0]
    x = F.conv2d(x, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], stride=ps)
    x = x.flatten(2).transpose(1, 2)  # B, T, D
    cls = w["cls_token"].expand(B, -1, -1)
    x = torch.cat([cls, x], dim=1)
    x = x + interpolate_pos_embed(w["pos_embed"], patch_h, pw)
    depth = 0
    while f"blocks.{depth}.norm1.weight" in w:
        depth += 1
    heads = Dm // 64
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (Dm,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-6)
        qkv = F.linear(h, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"])
        T = qkv.shape[1]
        qkv = qkv.reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax((q @ k.transpose(-1, -2)) * (64 ** -0.5), dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, T, Dm)
        o = F.linear(o, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        x = x + w[p + "ls1.gamma"] * o
        h = F.layer_norm(x, (Dm,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-6)
        h = F.linear(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"])
        h = F.gelu(h)
        h = F.linear(h, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"])
        x = x + w[p + "ls2.gamma"] * h
    x = F.layer_norm(x, (Dm,), w["norm.weight"], w["norm.bias"], 1e-6)
    x = x[:, 1:, :]  # drop cls: x_norm_patchtokens
    x = F.layer_norm(x, (Dm,), w["channel_norm.weight"], w["channel_norm.bias"], 1e-5)
    return x.reshape(B, patch_h, pw, Dm).numpy()


def interpolate_pos_embed(pos_embed, h: int, w: int):
    """dinov2 interpolate_pos_encoding: bicubic resize of the 37x37 patch grid to h x w
    (scale_factor form with the +0.1 offset as in facebookresearch/dinov2 vision_transformer.py)."""
    import torch
    import torch.nn.functional as F

    pe = torch.as_tensor(pos_embed, dtype=torch.float32)
    n = pe.shape[1] - 1
    m = int(round(n ** 0.5))
    cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
    if h == m and w == m:
        return pe
    dim = pe.shape[-1]
    h0, w0 = h + 0.1, w + 0.1
    patch_pe = F.interpolate(patch_pe.reshape(1, m, m, dim).permute(0, 3, 1, 2),
                             scale_factor=(h0 / m, w0 / m), mode="bicubic", align_corners=False)
    assert patch_pe.shape[-2] == h and patch_pe.shape[-1] == w
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, h * w, dim)
    return torch.cat([cls_pe, patch_pe], dim=1)


# ----------------------------------------------------------------------------- F2 ICP
def voxel_grid_csr(points: np.ndarray, voxel_size: float):
    """Sorted-key CSR of a point set (keys as in vfm_oracle.c:orc_voxel_key); points of a voxel keep
    their order (the insertion order of VoxelHashMap::AddPoints, VHM:733-770)."""
    pts = np.ascontiguousarray(points[:, :3], dtype=np.float64)
    v = np.trunc(pts / voxel_size).astype(np.int64) + (1 << 20)
    keys = (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    uniq, first = np.unique(ks, return_index=True)
    start = np.r_[first, len(ks)].astype(np.int32)
    return np.ascontiguousarray(uniq), start, np.ascontiguousarray(pts[order])


def se3_exp(dx: np.ndarray) -> np.ndarray:
    """Sophus::SE3d::exp (tangent = [upsilon (translation), omega (rotation)]) as a 4x4 matrix."""
    ups, om = np.asarray(dx[:3], np.float64), np.asarray(dx[3:], np.float64)
    th = float(np.linalg.norm(om))
    Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + Om
        V = np.eye(3) + 0.5 * Om
    else:
        R = np.eye(3) + np.sin(th) / th * Om + (1 - np.cos(th)) / th ** 2 * (Om @ Om)
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Om + (th - np.sin(th)) / th ** 3 * (Om @ Om)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ ups
    return T


def register_frame(points: np.ndarray, map_points: np.ndarray, voxel_size: float, initial_guess: np.ndarray,
                   max_correspondance_distance: float, kernel: float, max_iter: int = 1000,
                   return_history: bool = False):
    """kiss_icp RegisterFrame (Registration.cpp:145-195) on an explicit map point set."""
    keys, start, pts = voxel_grid_csr(map_points, voxel_size)
    src = np.ascontiguousarray(points[:, :3], dtype=np.float64)
    source = np.empty_like(src)
    T0 = np.ascontiguousarray(initial_guess, dtype=np.float64)
    lib().orc_transform_xyz(_p(src, _f64p), C.c_int64(len(src)), _p(T0, _f64p), _p(source, _f64p))
    T_icp = np.eye(4)
    hist = []
    n = len(source)
    tgt = np.empty_like(source)
    valid = np.empty(n, dtype=np.uint8)
    out = np.empty(43, dtype=np.float64)
    for _ in range(max_iter):
        lib().orc_icp_nearest(_p(source, _f64p), C.c_int64(n), _p(keys, _i64p), _p(start, _i32p), _p(pts, _f64p),
                              C.c_int32(len(keys)), C.c_double(voxel_size), C.c_double(max_correspondance_distance),
                              _p(tgt, _f64p), _p(valid, _u8p))
        lib().orc_icp_system(_p(source, _f64p), _p(tgt, _f64p), _p(valid, _u8p), C.c_int64(n), C.c_double(kernel),
                             _p(out, _f64p))
        if out[42] == 0:
            break
        JTJ, JTr = out[:36].reshape(6, 6), out[36:42]
        dx = np.linalg.solve(JTJ, -JTr)
        est = se3_exp(dx)
        new = np.empty_like(source)
        lib().orc_transform_xyz(_p(source, _f64p), C.c_int64(n), _p(np.ascontiguousarray(est), _f64p), _p(new, _f64p))
        source = new
        T_icp = est @ T_icp
        hist.append((out.copy(), dx.copy()))
        if np.linalg.norm(dx) < 1e-4:
            break
    T = T_icp @ T0
    return (T, hist) if return_history else T


def register_frame_xd(points: np.ndarray, map_rows: np.ndarray, voxel_size: float, initial_guess: np.ndarray,
                      max_correspondance_distance: float, kernel: float, max_iter: int = 1000, return_history: bool = False):
    """kiss_icp RegisterFrame(std::vector<Eigen::VectorXd> ...) (Registration.cpp:384-423) on an explicit map row set (xyz + descriptors of
    any width, the rows of map_x_ in insertion order): the 3-D loop with VoxelHashMap::GetCorrespondences(VectorXdVector)
    (VoxelHashMap.cpp:321-448, orc_icp_nearest_desc) as its search.  An iteration without correspondences ends the loop (the reference
    solves an all-zero system there: dx = 0, below the threshold)."""
    pts_all = np.asarray(points, dtype=np.float64)
    rows = np.asarray(map_rows, dtype=np.float64)
    f = pts_all.shape[1] - 3
    assert rows.shape[1] == pts_all.shape[1] and f >= 1
    v = np.trunc(rows[:, :3] / voxel_size).astype(np.int64) + (1 << 20)
    k = (v[:, 0] << 42) | (v[:, 1] << 21) | v[:, 2]
    order = np.argsort(k, kind="stable")
    keys, first = np.unique(k[order], return_index=True)
    keys = np.ascontiguousarray(keys)
    start = np.r_[first, len(order)].astype(np.int32)
    mpts = np.ascontiguousarray(rows[order, :3])
    mdesc = np.ascontiguousarray(rows[order, 3:])
    sdesc = np.ascontiguousarray(pts_all[:, 3:])
    mnorm, snorm = np.empty(len(mdesc)), np.empty(len(sdesc))
    mhas, shas = np.empty(len(mdesc), np.uint8), np.empty(len(sdesc), np.uint8)
    lib().orc_icp_desc_stats(_p(mdesc, _f64p), C.c_int64(len(mdesc)), C.c_int32(f), _p(mnorm, _f64p), _p(mhas, _u8p))
    lib().orc_icp_desc_stats(_p(sdesc, _f64p), C.c_int64(len(sdesc)), C.c_int32(f), _p(snorm, _f64p), _p(shas, _u8p))
    T0 = np.ascontiguousarray(initial_guess, dtype=np.float64)
    source = _transform_rows(np.ascontiguousarray(pts_all[:, :3]), T0)
    n = len(source)
    tgt = np.empty_like(source)
    valid = np.empty(n, dtype=np.uint8)
    out = np.empty(43, dtype=np.float64)
    T_icp = np.eye(4)
    hist = []
    for _ in range(max_iter):
        lib().orc_icp_nearest_desc(_p(source, _f64p), C.c_int64(n), _p(sdesc, _f64p), _p(snorm, _f64p), _p(shas, _u8p), C.c_int32(f),
                                   _p(keys, _i64p), _p(start, _i32p), _p(mpts, _f64p), _p(mdesc, _f64p), _p(mnorm, _f64p), _p(mhas, _u8p),
                                   C.c_int32(len(keys)), C.c_double(voxel_size), C.c_double(max_correspondance_distance),
                                   _p(tgt, _f64p), _p(valid, _u8p))
        lib().orc_icp_system(_p(source, _f64p), _p(tgt, _f64p), _p(valid, _u8p), C.c_int64(n), C.c_double(kernel), _p(out, _f64p))
        if out[42] == 0:
            break
        dx = np.linalg.solve(out[:36].reshape(6, 6), -out[36:42])
        est = se3_exp(dx)
        source = _transform_rows(source, est)
        T_icp = est @ T_icp
        hist.append((out.copy(), dx.copy(), tgt.copy(), valid.copy()))
        if np.linalg.norm(dx) < 1e-4:
            break
    T = T_icp @ T0
    return (T, hist) if return_history else T


def _transform_rows(xyz: np.ndarray, T: np.ndarray) -> np.ndarray:
    out = np.empty_like(xyz)
    lib().orc_transform_xyz(_p(np.ascontiguousarray(xyz), _f64p), C.c_int64(len(xyz)), _p(np.ascontiguousarray(T, dtype=np.float64), _f64p),
                            _p(out, _f64p))
    return out


def median_like_the_reference(v: np.ndarray) -> float:
    """Registration.cpp:297-309: nth_element at n = size / 2; for an even size the mean of that element and the largest of the
    elements before it -- the two middle order statistics."""
    s = np.sort(np.asarray(v, dtype=np.float64))
    n = len(s) // 2
    return float(s[n]) if len(s) & 1 else float((s[n] + s[n - 1]) / 2)


def register_frame_nd(points: np.ndarray, map_points_n: np.ndarray, voxel_size: float, initial_guess: np.ndarray,
                      max_correspondance_distance: float, kernel: float, min_cosine: float = 0.8, max_iter: int = 1000,
                      return_history: bool = False):
    """kiss_icp RegisterFrame(std::vector<VectorNd> ...) (Registration.cpp:197-382), the descriptor-seeded ICP, on an explicit map
    (``map_points_n``: the rows of VoxelHashMap::PointcloudN(), container order): scan moved by the initial guess (:207-208), 5 m
    voxel subset (first point per voxel, container order; the whole scan if fewer than 100 survive, :216-220), descriptor
    correspondences at cosine >= 0.8 (:229-230), Gauss-Newton on those pairs with median + 1.5 MAD pruning until the mean pair
    distance moves by less than 0.01 (:253-336), then the vanilla point-to-point loop on all points (:347-372, with the iteration
    counter carried over).  Returns (pose, src_, tgt_) -- the surviving descriptor pairs, the source side moved by every later
    update (:366) -- and the per-iteration history on request.  Choices where the reference leaves arithmetic open are the 3-D
    path's (register_frame): 4x4 matrix product for T * point, the fixed-tree normal equations, numpy's solve for LDLT; the
    vanilla loop stops when no pair is found (the reference prints and solves an empty system: dx = 0, the same exit)."""
    pts = np.asarray(points, dtype=np.float64)
    T0 = np.ascontiguousarray(initial_guess, dtype=np.float64)
    src_xyz = _transform_rows(np.ascontiguousarray(pts[:, :3]), T0)                       # :207-208
    source = np.c_[src_xyz, pts[:, 3:]]
    vox = voxel_down_sample(source, 5.0)                                                   # :216
    if len(vox) < 100:
        vox = source                                                                       # :217-220
    src_3d, tgt_3d, _, _, _ = get_vfm_correspondences(vox, np.asarray(map_points_n, dtype=np.float64), min_cosine)   # :229-230
    src_3d, tgt_3d = np.ascontiguousarray(src_3d[:, :3]), np.ascontiguousarray(tgt_3d[:, :3])

    def dists(a, b):
        d = a - b
        return np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    prev = float(np.sum(dists(src_3d, tgt_3d)) / len(src_3d)) if len(src_3d) else float("nan")   # :233-240 (in-order sum)
    source_3d = src_xyz
    T_icp = np.eye(4)
    hist = []
    out = np.empty(43, dtype=np.float64)
    j = 0
    while j < max_iter:                                                                    # :253
        if len(src_3d) == 0:
            break
        ones = np.ones(len(src_3d), dtype=np.uint8)
        lib().orc_icp_system(_p(src_3d, _f64p), _p(tgt_3d, _f64p), _p(ones, _u8p), C.c_int64(len(src_3d)), C.c_double(kernel), _p(out, _f64p))
        dx = np.linalg.solve(out[:36].reshape(6, 6), -out[36:42])
        est = se3_exp(dx)
        source_3d = _transform_rows(source_3d, est)                                        # :265-266
        src_3d = _transform_rows(src_3d, est)
        T_icp = est @ T_icp
        d = dists(src_3d, tgt_3d)
        mean = float(np.add.accumulate(d)[-1] / len(d))                                    # std::accumulate: in order
        median = median_like_the_reference(d)
        mad = median_like_the_reference(np.abs(d - median)) * 1.4826                       # :311-322
        keep = np.abs(d - median) < 1.5 * mad                                              # :326-331
        hist.append(("vfm", out.copy(), dx.copy(), int(keep.sum())))
        src_3d, tgt_3d = np.ascontiguousarray(src_3d[keep]), np.ascontiguousarray(tgt_3d[keep])
        if abs(prev - mean) < 0.01:                                                        # :332-334 (j is not incremented on break)
            break
        prev = mean
        j += 1
    src_, tgt_ = src_3d, tgt_3d
    keys, start, mpts = voxel_grid_csr(np.asarray(map_points_n)[:, :3], voxel_size)
    n = len(source_3d)
    tgt = np.empty_like(source_3d)
    valid = np.empty(n, dtype=np.uint8)
    while j < max_iter:                                                                    # :347
        lib().orc_icp_nearest(_p(source_3d, _f64p), C.c_int64(n), _p(keys, _i64p), _p(start, _i32p), _p(mpts, _f64p),
                              C.c_int32(len(keys)), C.c_double(voxel_size), C.c_double(max_correspondance_distance),
                              _p(tgt, _f64p), _p(valid, _u8p))
        lib().orc_icp_system(_p(source_3d, _f64p), _p(tgt, _f64p), _p(valid, _u8p), C.c_int64(n), C.c_double(kernel), _p(out, _f64p))
        if out[42] == 0:
            break
        dx = np.linalg.solve(out[:36].reshape(6, 6), -out[36:42])
        est = se3_exp(dx)
        source_3d = _transform_rows(source_3d, est)
        T_icp = est @ T_icp
        if len(src_):
            src_ = _transform_rows(src_, est)                                              # :366
        hist.append(("icp", out.copy(), dx.copy(), int(out[42])))
        if np.linalg.norm(dx) < 1e-4:
            break
        j += 1
    T = T_icp @ T0
    return (T, src_, tgt_, hist) if return_history else (T, src_, tgt_)


# ----------------------------------------------------------------------------- F4 evaluation harness
def build_local_map(map_poses, map_point_clouds, voxel_size: float = .25, n_descriptors: int = 384) -> np.ndarray:
    """RN:556-580: rows with descriptor sum <= 0 dropped, every cloud voxelised (container order), moved into the map
    frame, concatenated as float32, voxelised again (in two halves about the mean x above 1e6 rows)."""
    local = []
    for pose, pcl in zip(map_poses, map_point_clouds):
        pcl = pcl[np.sum(pcl[:, 3:], axis=1) > 0]
        pcl = voxel_down_sample(pcl, voxel_size).astype(pcl.dtype)
        local.append(transform_pcl(pcl, pose))
    m = np.concatenate(local, axis=0).astype(np.float32)
    if m.shape[0] > 1000000:
        mean_3d = np.mean(m[:, :3], axis=0)
        a = voxel_down_sample(m[m[:, 0] > mean_3d[0]], voxel_size).astype(m.dtype)
        b = voxel_down_sample(m[m[:, 0] <= mean_3d[0]], voxel_size).astype(m.dtype)
        m = np.concatenate([a, b], axis=0)
    else:
        m = voxel_down_sample(m, voxel_size).astype(m.dtype)
    return m[:, :3 + n_descriptors]


def ransac_registration_vfm(voxel_map: np.ndarray, raw_scan: np.ndarray, n_iter: int = 50000, seed: int = 42,
                            run_icp: bool = False, voxel_size: float = 1.0, max_points_per_voxel: int = 20,
                            sigma: float = 2.0, min_cosine: float = 0.8):
    """RegistrationNode.ransac_registration(voxel_map, raw_scan, 'vfm', run_icp) (RN:273-357 + 396-425) assembled
    from the oracle's pieces; returns (ransac_pose, icp_pose | None, correspondences)."""
    scan = voxel_down_sample(voxel_down_sample(raw_scan, voxel_size * 0.5), voxel_size * 1.0)     # RN:399-400
    mp = np.asarray(voxel_map, dtype=np.float64)[voxel_hash_map_points(voxel_map, voxel_size, max_points_per_voxel)]
    pcl = transform_pcl(scan, np.eye(4))                                                            # RN:408
    sub = voxel_down_sample(pcl, 5.0)                                                               # RN:414
    _, _, qi, mi, _ = get_vfm_correspondences(sub, mp, min_cosine)
    if len(qi) < 75:                                                                                # RN:420-423
        sub = voxel_down_sample(pcl, 1.0)
        _, _, qi, mi, _ = get_vfm_correspondences(sub, mp, min_cosine)
    # RN:288-309: rows of the correspondences inside the re-voxelised clouds (exact coordinates)
    key = lambda a: (np.ascontiguousarray(a, dtype=np.float64) + 0.0).view(np.dtype((np.void, 24))).reshape(-1)
    order = np.argsort(key(scan[:, :3]), kind="stable")
    pos = np.searchsorted(key(scan[:, :3])[order], key(sub[qi, :3]))
    src_rows = order[pos]
    corres = np.stack([src_rows, mi], 1).astype(np.int32)
    res = ransac_corr(scan[:, :3], mp[:, :3], corres, 10000.0, n_iter, seed=seed)
    pose = res.transformation
    if not run_icp:
        return pose, None, corres
    guess = orthogonalize_rotation(pose)                                                            # RN:331-336
    icp = register_frame(scan[:, :3], mp[:, :3], voxel_size, guess, 3 * sigma, sigma / 3)           # RN:338-344
    return guess, icp, corres


def success_rate(trans_errors, rot_errors, translation_threshold, rotation_threshold) -> float:
    """RN:1021-1025 / print_errors.py:8-13."""
    return float(np.mean((np.array(trans_errors) < translation_threshold) & (np.array(rot_errors) < rotation_threshold)))


def evaluate_scene(scene: dict, n_iter: int = 50000, run_icp: bool = True, n_descriptors: Optional[int] = None):
    """The VFM + RANSAC (+ ICP) branch of make_step for one scene (RN:556-593, 858-882, 943-951)."""
    n_desc = n_descriptors or scene["map_point_clouds"][0].shape[1] - 3
    local_map = build_local_map(scene["map_poses"], scene["map_point_clouds"], n_descriptors=n_desc)
    rot, trans, poses = {}, {}, []
    for gt_pose, cloud in zip(scene["scene_poses"], scene["scene_point_clouds"]):
        cloud = voxel_down_sample(cloud, .1).astype(cloud.dtype)                                    # RN:593
        cloud = transform_pcl(cloud, np.eye(4))                                                     # RN:863
        p0, p1, _ = ransac_registration_vfm(local_map, cloud, n_iter=n_iter, run_icp=run_icp)
        for k, v in (("vfm_ransac", p0), ("vfm_ransac_icp", p1)):
            if v is None:
                continue
            # RN:997-1011 literally (called as compute_errors(gt_pose, v, k) at RN:947)
            R, R_gt = np.asarray(gt_pose)[:3, :3], (v @ np.eye(4))[:3, :3]
            rre = float(np.rad2deg(abs(np.arccos(min(max(((R.T @ R_gt).trace() - 1) / 2, -1.0), 1.0)))))
            rte = float(np.linalg.norm(np.asarray(gt_pose)[:3, 3] - (v @ np.eye(4))[:3, 3]))
            trans.setdefault(k, []).append(rte)
            rot.setdefault(k, []).append(rre)
        poses.append((p0, p1))
    return dict(local_map=local_map, rot_errors=rot, trans_errors=trans, poses=poses)
