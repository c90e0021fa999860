"""Tensor-level wrappers over the C ABI (include/vfmreg.h).

Inputs and outputs are torch tensors resident on a ROCm device; PyTorch is used only for device
memory and the current HIP stream.  Nothing here synchronises the device.  There is no CPU
implementation: every function raises if the tensors are not on a GPU.
"""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import torch

from . import _lib

FAST = 0
EXACT = 1

PROJ_NCLT, PROJ_ROBOTCAR, PROJ_KITTI = 0, 1, 2


_tls = threading.local()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the ROCm device (no CPU path exists)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------- matching
def l2norm_rows_(x: torch.Tensor, inv_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In-place faiss::fvec_renorm_L2 per row (VoxelHashMap.cpp:474,480)."""
    _chk(x, torch.float32, "x")
    lib = _lib.load()
    _lib.check(lib.vfm_l2norm_rows_f32(x.data_ptr(), x.shape[0], x.shape[1], _ptr(inv_out), _stream()), "l2norm")
    return x


def match_ip_top1(q: torch.Tensor, b: torch.Tensor, prec: int = FAST,
                  ws: Optional[torch.Tensor] = None, gate: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Normalise + IndexFlatIP top-1 (VoxelHashMap.cpp:469-495). Returns (idx int64[N], sim fp32[N]).
    ``gate`` (the gated family of include/vfmreg.h): for a caller that keeps only matches with similarity >= gate
    (VoxelHashMap.cpp:501-511) -- queries that provably cannot reach it come back as (-1, -2.0); ``-inf`` resolves all."""
    _chk(q, torch.float32, "q")
    _chk(b, torch.float32, "b")
    if q.dim() != 2 or b.dim() != 2 or q.shape[1] != b.shape[1]:
        raise ValueError("Invalid shape")
    lib = _lib.load()
    n, d = q.shape
    m = b.shape[0]
    need = lib.vfm_match_ip_top1_workspace_bytes(n, m, d, prec)
    if ws is None or ws.numel() < need:
        ws = _ws(need, q.device)
    idx = torch.empty(n, dtype=torch.int64, device=q.device)
    sim = torch.empty(n, dtype=torch.float32, device=q.device)
    if gate is None:
        _lib.check(lib.vfm_match_ip_top1(q.data_ptr(), n, b.data_ptr(), m, d, prec, idx.data_ptr(), sim.data_ptr(),
                                         ws.data_ptr(), ws.numel(), _stream()), "match_ip_top1")
    else:
        _lib.check(lib.vfm_match_ip_top1_gated(q.data_ptr(), n, b.data_ptr(), m, d, prec, float(gate), idx.data_ptr(),
                                               sim.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "match_ip_top1")
    return idx, sim


class PreparedRows:
    """IndexFlatIP.add (VoxelHashMap.cpp:486-487): 1/|row| + fp16 fragment tiles of the rows."""

    def __init__(self, x: torch.Tensor):
        _chk(x, torch.float32, "x")
        lib = _lib.load()
        self.x = x
        self.rows, self.d = x.shape
        self.buf = _ws(lib.vfm_match_prepared_bytes(self.rows, self.d), x.device)
        self.refresh()

    def refresh(self) -> None:
        lib = _lib.load()
        _lib.check(lib.vfm_match_prepare(self.x.data_ptr(), self.rows, self.d, self.buf.data_ptr(), _stream()),
                   "match_prepare")


def match_search(q: PreparedRows, b: PreparedRows, idx: Optional[torch.Tensor] = None,
                 sim: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None):
    lib = _lib.load()
    if q.d != b.d:
        raise ValueError("Invalid shape")
    need = lib.vfm_match_search_workspace_bytes(q.rows, b.rows, q.d)
    if ws is None or ws.numel() < need:
        ws = _ws(need, q.x.device)
    if idx is None:
        idx = torch.empty(q.rows, dtype=torch.int64, device=q.x.device)
    if sim is None:
        sim = torch.empty(q.rows, dtype=torch.float32, device=q.x.device)
    _lib.check(lib.vfm_match_search_prepared(q.x.data_ptr(), q.buf.data_ptr(), q.rows, b.x.data_ptr(),
                                             b.buf.data_ptr(), b.rows, q.d, idx.data_ptr(), sim.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _stream()), "match_search")
    return idx, sim


def gated_split_ok(d: int) -> bool:
    """Widths for which the gated family's split calls exist (the int8 coarse pass: d = 256 ... 768 in steps of 128)."""
    return d % 128 == 0 and 256 <= d <= 768


def match_search_gated(q: PreparedRows, b: PreparedRows, gate: float, idx: Optional[torch.Tensor] = None,
                       sim: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None, records: Optional[int] = None,
                       rescans_out: Optional[torch.Tensor] = None):
    """vfm_match_search_coarse_gated + vfm_match_search_finish_gated on prepared operands: what vfm_match_ip_top1_gated does after
    preparing both -- for a map that is prepared once and searched by many scans.  Same answers (idx -1 / sim -2.0 for queries that
    provably cannot reach ``gate``).  ``records``: the record kind of the coarse pass (include/vfmreg.h VFM_RECORDS_*; None = the
    default, best-score records); ``rescans_out``: a pinned 1-element int32 tensor that receives the search's load figure
    (vfm_match_search_rescans_async: candidate chunks rescanned) once the stream has passed this call."""
    lib = _lib.load()
    if q.d != b.d or not gated_split_ok(q.d):
        raise ValueError("Invalid shape")
    dev = q.x.device
    need = lib.vfm_match_search_workspace_bytes(q.rows, b.rows, q.d)
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    if idx is None:
        idx = torch.empty(q.rows, dtype=torch.int64, device=dev)
    if sim is None:
        sim = torch.empty(q.rows, dtype=torch.float32, device=dev)
    st = _stream()
    if records is None:
        _lib.check(lib.vfm_match_search_coarse_gated(q.buf.data_ptr(), q.rows, b.buf.data_ptr(), b.rows, q.d, ws.data_ptr(), ws.numel(), st),
                   "search(coarse)")
        _lib.check(lib.vfm_match_search_finish_gated(q.x.data_ptr(), q.buf.data_ptr(), q.rows, b.x.data_ptr(), b.buf.data_ptr(), b.rows, q.d,
                                                     idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), float(gate), st), "search(finish)")
    else:
        _lib.check(lib.vfm_match_search_coarse_gated_g(q.buf.data_ptr(), q.rows, b.buf.data_ptr(), b.rows, q.d, ws.data_ptr(), ws.numel(),
                                                       int(records), float(gate), st), "search(coarse)")
        _lib.check(lib.vfm_match_search_finish_gated_r(q.x.data_ptr(), q.buf.data_ptr(), q.rows, b.x.data_ptr(), b.buf.data_ptr(), b.rows, q.d,
                                                       idx.data_ptr(), sim.data_ptr(), ws.data_ptr(), ws.numel(), float(gate), int(records), st),
                   "search(finish)")
    if rescans_out is not None:
        _lib.check(lib.vfm_match_search_rescans_async(ws.data_ptr(), q.rows, b.rows, rescans_out.data_ptr(), st), "rescans")
    return idx, sim


def match_probe_half(q: PreparedRows, b: PreparedRows, gate: float, out: torch.Tensor, ws: Optional[torch.Tensor] = None) -> None:
    """vfm_match_search_probe_half: the (query, chunk) pairs the half-width coarse pass would leave for this pair, to the pinned
    1-element int32 tensor ``out`` (asynchronously on the current stream)."""
    lib = _lib.load()
    need = lib.vfm_match_search_workspace_bytes(q.rows, b.rows, q.d)
    if ws is None or ws.numel() < need:
        ws = _ws(need, q.x.device)
    _lib.check(lib.vfm_match_search_probe_half(q.buf.data_ptr(), q.rows, b.buf.data_ptr(), b.rows, q.d, ws.data_ptr(), ws.numel(),
                                               float(gate), out.data_ptr(), _stream()), "probe_half")


def threshold_compact(sim: torch.Tensor, idx: Optional[torch.Tensor], thr: float,
                      q_xyz: Optional[torch.Tensor] = None, b_xyz: Optional[torch.Tensor] = None,
                      want_corres: bool = True):
    """VoxelHashMap.cpp:501-511 + 587-600. Returns dict(keep, count, corres, src, tgt); `count` is a
    1-element int64 device tensor (no host sync); the arrays are sized N, valid up to count."""
    _chk(sim, torch.float32, "sim")
    n = sim.shape[0]
    dev = sim.device
    lib = _lib.load()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    corres = torch.empty((n, 2), dtype=torch.int32, device=dev) if (want_corres and idx is not None) else None
    src = torch.empty((n, 3), dtype=torch.float64, device=dev) if q_xyz is not None else None
    tgt = torch.empty((n, 3), dtype=torch.float64, device=dev) if (b_xyz is not None and idx is not None) else None
    if idx is not None:
        _chk(idx, torch.int64, "idx")
    if q_xyz is not None:
        _chk(q_xyz, torch.float64, "q_xyz")
    if b_xyz is not None:
        _chk(b_xyz, torch.float64, "b_xyz")
    _lib.check(lib.vfm_threshold_compact(sim.data_ptr(), _ptr(idx), n, float(thr), keep.data_ptr(), count.data_ptr(),
                                         _ptr(corres), _ptr(q_xyz), _ptr(b_xyz), _ptr(src), _ptr(tgt), _stream()),
               "threshold_compact")
    return dict(keep=keep, count=count, corres=corres, src=src, tgt=tgt)


def match_mutual_l2(a: torch.Tensor, b: torch.Tensor, mutual: bool = True, prec: int = FAST):
    """Exact Euclidean 1-NN a->b (and b->a) (registration_node.py:485-496).  FAST and EXACT return the same
    indices and distances; FAST runs the all-pairs part on the matrix cores."""
    _chk(a, torch.float32, "a")
    _chk(b, torch.float32, "b")
    if a.shape[1] != b.shape[1]:
        raise ValueError("Invalid shape")
    lib = _lib.load()
    n, m, d = a.shape[0], b.shape[0], a.shape[1]
    nn_ab = torch.empty(n, dtype=torch.int64, device=a.device)
    d2 = torch.empty(n, dtype=torch.float64, device=a.device)
    nn_ba = torch.empty(m, dtype=torch.int64, device=a.device) if mutual else None
    ws = torch.empty(lib.vfm_match_mutual_l2_workspace_bytes(n, m, d, prec, int(mutual)), dtype=torch.uint8, device=a.device)
    _lib.check(lib.vfm_match_mutual_l2(a.data_ptr(), n, b.data_ptr(), m, d, prec, nn_ab.data_ptr(), d2.data_ptr(),
                                       _ptr(nn_ba), ws.data_ptr(), ws.numel(), _stream()), "match_mutual_l2")
    return nn_ab, d2, nn_ba


def match_mutual_pairs(a: torch.Tensor, b: torch.Tensor, want_nn: bool = False):
    """find_correspondences(mutual_filter=True) in one call (registration_node.py:482-538): (idx0, idx1, count[, nn_ab, d2_ab]);
    idx0 / idx1 hold ``count`` valid pairs in ascending idx0 (device tensors, count is a 1-element int64 tensor: no host sync)."""
    _chk(a, torch.float32, "a")
    _chk(b, torch.float32, "b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("Invalid shape")
    lib = _lib.load()
    n, m, d = a.shape[0], b.shape[0], a.shape[1]
    dev = a.device
    idx0 = torch.empty(n, dtype=torch.int64, device=dev)
    idx1 = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    nn_ab = torch.empty(n, dtype=torch.int64, device=dev) if want_nn else None
    d2 = torch.empty(n, dtype=torch.float64, device=dev) if want_nn else None
    ws = _ws(lib.vfm_match_mutual_pairs_workspace_bytes(n, m, d), dev)
    _lib.check(lib.vfm_match_mutual_pairs(a.data_ptr(), n, b.data_ptr(), m, d, idx0.data_ptr(), idx1.data_ptr(), count.data_ptr(),
                                          _ptr(nn_ab), _ptr(d2), ws.data_ptr(), ws.numel(), _stream()), "match_mutual_pairs")
    return (idx0, idx1, count, nn_ab, d2) if want_nn else (idx0, idx1, count)


# --------------------------------------------------------------------------------------- RANSAC
def ransac_corr(src: torch.Tensor, tgt: torch.Tensor, corres: torch.Tensor, max_dist: float, n_iter: int,
                seed: int = 42, count: Optional[torch.Tensor] = None, want_mask: bool = True,
                ws: Optional[torch.Tensor] = None, out: Optional[dict] = None, check_bounds: bool = False):
    """registration_ransac_based_on_correspondence (registration_node.py:319-327).  `count`
    (1-element int64 device tensor) gives the number of valid rows of `corres` without a host sync.  ``check_bounds``: the kernel that
    gathers the point pairs tests every index against the clouds' lengths (vfm_ransac_corr_bounded); ``out["bad"]`` (int32[1]) is 1
    if one was out of range -- such an entry is read as row 0 and the caller discards the result."""
    _chk(src, torch.float64, "src")
    _chk(tgt, torch.float64, "tgt")
    _chk(corres, torch.int32, "corres")
    lib = _lib.load()
    dev = src.device
    c_max = corres.shape[0]
    need = lib.vfm_ransac_workspace_bytes(c_max, n_iter)
    if ws is None or ws.numel() < need:
        ws = _ws(need, dev)
    if out is None:
        out = dict(T=torch.empty((4, 4), dtype=torch.float64, device=dev),
                   fitness=torch.empty(1, dtype=torch.float64, device=dev),
                   rmse=torch.empty(1, dtype=torch.float64, device=dev),
                   best_hyp=torch.empty(1, dtype=torch.int32, device=dev),
                   mask=torch.empty(max(c_max, 1), dtype=torch.uint8, device=dev) if want_mask else None)
    if count is not None:
        _chk(count, torch.int64, "count")
    if check_bounds:
        out["bad"] = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.vfm_ransac_corr_bounded(src.data_ptr(), src.shape[0], tgt.data_ptr(), tgt.shape[0], corres.data_ptr(), _ptr(count), c_max,
                                               float(max_dist), int(n_iter), int(seed) & 0xFFFFFFFFFFFFFFFF, out["T"].data_ptr(),
                                               out["fitness"].data_ptr(), out["rmse"].data_ptr(), _ptr(out.get("mask")),
                                               out["best_hyp"].data_ptr(), out["bad"].data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
                   "ransac_corr")
        return out
    _lib.check(lib.vfm_ransac_corr(src.data_ptr(), tgt.data_ptr(), corres.data_ptr(), _ptr(count), c_max,
                                   float(max_dist), int(n_iter), int(seed) & 0xFFFFFFFFFFFFFFFF, out["T"].data_ptr(),
                                   out["fitness"].data_ptr(), out["rmse"].data_ptr(), _ptr(out.get("mask")),
                                   out["best_hyp"].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ransac_corr")
    return out


def kabsch_batched(A: torch.Tensor, B: torch.Tensor, w: Optional[torch.Tensor] = None, denom_eps: float = 0.0):
    """Eigen::umeyama (no scaling) / pointdsc rigid_transform_3d, batched: A, B [b, n, 3] fp64."""
    _chk(A, torch.float64, "A")
    _chk(B, torch.float64, "B")
    if w is not None:
        _chk(w, torch.float64, "w")
    lib = _lib.load()
    b, n, _ = A.shape
    T = torch.empty((b, 4, 4), dtype=torch.float64, device=A.device)
    valid = torch.empty(b, dtype=torch.int32, device=A.device)
    _lib.check(lib.vfm_kabsch_batched(A.data_ptr(), B.data_ptr(), _ptr(w), b, n, float(denom_eps), T.data_ptr(),
                                      valid.data_ptr(), _stream()), "kabsch_batched")
    return T, valid


# ----------------------------------------------------------------------------------- projection
def project_pinhole(mode: int, pcl4xn: torch.Tensor, mats, fc=None, subsample: float = 1.0, win=None,
                    image: Optional[torch.Tensor] = None, H: int = 0, W: int = 0):
    """Dataset.project_pcl_to_image. pcl4xn: [4, N] fp64 device tensor. mats: up to three host
    matrices (see include/vfmreg.h). Returns (u int32[N], v int32[N], idx int64[N], count int64[1])."""
    import ctypes as C
    _chk(pcl4xn, torch.float64, "pcl")
    if pcl4xn.dim() != 2 or pcl4xn.shape[0] != 4:
        raise ValueError("Invalid shape")
    lib = _lib.load()
    n = pcl4xn.shape[1]
    dev = pcl4xn.device
    flat = [0.0] * 48
    for k, m in enumerate(list(mats)[:3]):
        vals = [float(x) for x in (m.reshape(-1).tolist() if hasattr(m, "reshape") else m)]
        flat[16 * k:16 * k + len(vals)] = vals
    mats_c = (C.c_double * 48)(*flat)
    fc_c = (C.c_double * 4)(*([float(x) for x in fc] if fc is not None else [0.0] * 4))
    win_c = (C.c_int64 * 4)(*([int(x) for x in win] if win is not None else [0] * 4))
    if image is not None:
        _chk(image, torch.uint8, "image")
        H, W = image.shape[0], image.shape[1]
    u = torch.empty(n, dtype=torch.int32, device=dev)
    v = torch.empty(n, dtype=torch.int32, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    ws = _ws(lib.vfm_project_workspace_bytes(n), dev)
    _lib.check(lib.vfm_project_pinhole_f64(mode, pcl4xn.data_ptr(), n, C.cast(mats_c, C.c_void_p),
                                           C.cast(fc_c, C.c_void_p), float(subsample), C.cast(win_c, C.c_void_p),
                                           _ptr(image), int(H), int(W), u.data_ptr(), v.data_ptr(), idx.data_ptr(),
                                           count.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "project")
    return u, v, idx, count


LIFT_MAX_CAMS = 6


class LiftPlan:
    """The camera records of ``lift_multicam`` marshalled once (mode, matrices, image / grid pointers): a caller that lifts
    scan after scan with the same rig -- prepare_scenes.py:50-107 walks a sequence -- pays the Python-side packing once, and a
    call is one kernel launch.  The tensors named in ``cams`` are kept alive by the plan; their CONTENTS may change between
    calls (new images, new patch grids in the same buffers)."""

    def __init__(self, cams: list, C: int):
        if not 1 <= len(cams) <= LIFT_MAX_CAMS:
            raise ValueError("Invalid shape")
        self.C = int(C)
        self.arr = (_lib.LiftCamera * len(cams))()
        self.keep = []
        for k, c in enumerate(cams):
            g = c["grid"]
            _chk(g, torch.float32, "grid")
            if g.shape[-1] != self.C:
                raise ValueError("Invalid shape")
            a = self.arr[k]
            a.mode = int(c["mode"])
            flat = [0.0] * 48
            for j, m in enumerate(list(c["mats"])[:3]):
                vals = [float(x) for x in (m.reshape(-1).tolist() if hasattr(m, "reshape") else m)]
                flat[16 * j:16 * j + len(vals)] = vals
            a.mats[:] = flat
            a.fc[:] = [float(x) for x in c["fc"]] if c.get("fc") is not None else [0.0] * 4
            a.subsample = float(c.get("subsample", 1.0))
            a.win[:] = [int(x) for x in c["win"]] if c.get("win") is not None else [0] * 4
            a.H, a.W = int(c["H"]), int(c["W"])
            for name in ("proj_image", "raw_image"):
                t = c.get(name)
                if t is not None:
                    _chk(t, torch.uint8, name)
                    self.keep.append(t)
                setattr(a, name, _ptr(t))
            self.keep.append(g)
            a.grid = g.data_ptr()
            a.gh, a.gw = int(g.shape[0]), int(g.shape[1])
            a.Hup, a.Wup, a.rot_mode = int(c["Hup"]), int(c["Wup"]), int(c.get("rot_mode", 0))
        import ctypes as C_
        self._ptr = C_.cast(self.arr, C_.c_void_p)
        self._lib = _lib.load()

    def __call__(self, pcl4xn: torch.Tensor, desc: torch.Tensor, filled: torch.Tensor) -> None:
        _chk(pcl4xn, torch.float64, "pcl")
        _chk(desc, torch.float32, "desc")
        _chk(filled, torch.uint8, "filled")
        if pcl4xn.dim() != 2 or pcl4xn.shape[0] != 4 or desc.shape[1] != self.C:
            raise ValueError("Invalid shape")
        _lib.check(self._lib.vfm_lift_multicam(pcl4xn.data_ptr(), pcl4xn.shape[1], len(self.arr), self._ptr, self.C,
                                               desc.data_ptr(), filled.data_ptr(), _stream()), "lift_multicam")


def lift_multicam(pcl4xn: torch.Tensor, cams: list, desc: torch.Tensor, filled: torch.Tensor) -> None:
    """create_descriptors (prepare_scenes.py:50-107) for all cameras in one launch.  ``cams``: list (priority
    order) of dicts with the projection parameters of ``project_pinhole`` (mode, mats, fc, subsample, win, H, W,
    proj_image) and of ``gather_bilinear`` (grid [gh, gw, C], Hup, Wup, rot_mode, raw_image).  Every row of ``desc`` is
    written (zeros for points no camera sees / black pixels); ``filled`` receives 1 for every point some camera saw."""
    LiftPlan(cams, desc.shape[1])(pcl4xn, desc, filled)


def gather_bilinear(grid: torch.Tensor, Hup: int, Wup: int, rot_mode: int, image: Optional[torch.Tensor],
                    u: torch.Tensor, v: torch.Tensor, idx: torch.Tensor, count: Optional[torch.Tensor],
                    desc: torch.Tensor, filled: torch.Tensor) -> None:
    """One camera of create_descriptors (prepare_scenes.py:57-104) fused with the bilinear upsample
    of image_features.py:104-108; call per camera in priority order."""
    _chk(grid, torch.float32, "grid")
    _chk(u, torch.int32, "u")
    _chk(v, torch.int32, "v")
    _chk(idx, torch.int64, "idx")
    _chk(desc, torch.float32, "desc")
    _chk(filled, torch.uint8, "filled")
    if image is not None:
        _chk(image, torch.uint8, "image")
    gh, gw, Cc = grid.shape
    lib = _lib.load()
    _lib.check(lib.vfm_gather_bilinear_patchgrid(grid.data_ptr(), gh, gw, Cc, int(Hup), int(Wup), int(rot_mode),
                                                 _ptr(image), u.data_ptr(), v.data_ptr(), idx.data_ptr(), _ptr(count),
                                                 u.shape[0], desc.data_ptr(), filled.data_ptr(), _stream()), "gather")


def transform_xyz(xyz: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """transform_pcl's coordinate part (vfm_reg/utils.py:47-54): fp64 [N,3] x [4,4]."""
    _chk(xyz, torch.float64, "xyz")
    _chk(T, torch.float64, "T")
    out = torch.empty_like(xyz)
    lib = _lib.load()
    _lib.check(lib.vfm_transform_xyz_f64(xyz.data_ptr(), xyz.shape[0], T.data_ptr(), out.data_ptr(), _stream()),
               "transform_xyz")
    return out


# ------------------------------------------------------------------------------------ voxel maps
def voxel_robin_level(pts: torch.Tensor, voxel_size: float, idx: Optional[torch.Tensor] = None, n_dev: Optional[torch.Tensor] = None,
                      n_max: Optional[int] = None, T: Optional[torch.Tensor] = None, hash_mul: int = 19349663, want_local: bool = False):
    """One level of a chain of VoxelDownsample()s, enqueued without a read-back (vfm_voxel_robin_level): the points ``pts[idx[:n_dev]]``
    (``idx`` None: ``pts`` itself), moved by the pose ``T`` first when given.  Returns ``dict(keep, local, count, info)`` of device tensors:
    ``keep`` the survivors in the container's order as rows of ``pts`` (the next level's ``idx``), ``local`` their positions in this
    level's input (``want_local``), ``count`` int64[1], ``info`` int64[8] = {buckets, voxels or -1, probe distance, wrapped, -, done}."""
    _chk(pts, torch.float64, "pts")
    lib = _lib.load()
    stride = pts.shape[1]
    n_max = int(n_max if n_max is not None else (idx.shape[0] if idx is not None else pts.shape[0]))
    dev = pts.device
    keep = torch.empty(n_max, dtype=torch.int64, device=dev)
    local = torch.empty(n_max, dtype=torch.int64, device=dev) if want_local else None
    count = torch.empty(1, dtype=torch.int64, device=dev)
    info = torch.empty(8, dtype=torch.int64, device=dev)
    ws = _ws(lib.vfm_voxel_robin_workspace_bytes(n_max), dev)
    if idx is not None:
        _chk(idx, torch.int64, "idx")
    if T is not None:
        _chk(T, torch.float64, "T")
    _lib.check(lib.vfm_voxel_robin_level(pts.data_ptr(), stride, _ptr(idx), n_max, _ptr(n_dev), _ptr(T), float(voxel_size), int(hash_mul),
                                         keep.data_ptr(), _ptr(local), count.data_ptr(), info.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "voxel_robin_level")
    return dict(keep=keep, local=local, count=count, info=info, ws=ws)



def voxel_first(xyz: torch.Tensor, voxel_size: float, max_per_voxel: int = 1) -> torch.Tensor:
    """Indices (ascending) of the points that are among the first `max_per_voxel` of their voxel:
    VoxelDownsample (Preprocessing.cpp:50-137) for 1, VoxelHashMap::AddPoints' cap otherwise."""
    _chk(xyz, torch.float64, "xyz")
    lib = _lib.load()
    n, stride = xyz.shape
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=xyz.device)
    count = torch.empty(1, dtype=torch.int64, device=xyz.device)
    ws = _ws(lib.vfm_voxel_first_workspace_bytes(n), xyz.device)
    _lib.check(lib.vfm_voxel_first(xyz.data_ptr(), n, stride, float(voxel_size), int(max_per_voxel), keep.data_ptr(),
                                   count.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "voxel_first")
    return keep[:int(count.item())]


HASH_DOWNSAMPLE = 19349663  # VoxelHash of Preprocessing.cpp:44
HASH_MAP = 19349669         # VoxelHash of VoxelHashMap.hpp:75


def voxel_robin(xyz: torch.Tensor, voxel_size: float, max_per_voxel: int = 1, reserve: bool = True,
                hash_mul: int = HASH_DOWNSAMPLE, return_info: bool = False):
    """The survivors of ``voxel_first`` in the order the reference emits them (tsl::robin_map iteration
    order): ``reserve=True, HASH_DOWNSAMPLE`` = VoxelDownsample (Preprocessing.cpp:50-69);
    ``reserve=False, HASH_MAP`` = a fresh VoxelHashMap after AddPoints, as Pointcloud*() walk it."""
    _chk(xyz, torch.float64, "xyz")
    lib = _lib.load()
    n, stride = xyz.shape
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=xyz.device)
    count = torch.empty(1, dtype=torch.int64, device=xyz.device)
    info = getattr(_tls, "voxel_info", None)     # page-locked: the entry point's one read-back lands in it by DMA
    if info is None:
        info = _tls.voxel_info = torch.zeros(4, dtype=torch.int64).pin_memory()
    ws = _ws(lib.vfm_voxel_robin_workspace_bytes(n), xyz.device)
    _lib.check(lib.vfm_voxel_robin(xyz.data_ptr(), n, stride, float(voxel_size), int(max_per_voxel), int(hash_mul),
                                   n if reserve else -1, keep.data_ptr(), count.data_ptr(), info.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _stream()), "voxel_robin")
    # (the entry point synchronises and reports the number of voxels: with one point per voxel that IS the count -- no second read-back)
    vals = info.tolist()
    out = keep[:vals[1] if max_per_voxel == 1 else int(count.item())]
    return (out, vals) if return_info else out
