"""ctypes loader for libvfmreg_hip.so (the C ABI declared in include/vfmreg.h).

The product path has NO CPU fallback: if the HIP library is missing this module raises, and every
op in ``vfmreg.ops`` refuses to run without a ROCm device.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libvfmreg_hip.so"
_lib = None

c_i64 = C.c_int64
c_vp = C.c_void_p


class VitConfig(C.Structure):
    _fields_ = [("dim", C.c_int), ("depth", C.c_int), ("heads", C.c_int), ("mlp_dim", C.c_int),
                ("patch", C.c_int), ("patch_h", C.c_int), ("patch_w", C.c_int)]


class LiftCamera(C.Structure):  # vfm_lift_camera
    _fields_ = [("mode", C.c_int), ("mats", C.c_double * 48), ("fc", C.c_double * 4), ("subsample", C.c_double),
                ("win", C.c_int64 * 4), ("H", C.c_int64), ("W", C.c_int64), ("proj_image", C.c_void_p),
                ("grid", C.c_void_p), ("raw_image", C.c_void_p), ("gh", C.c_int), ("gw", C.c_int), ("Hup", C.c_int),
                ("Wup", C.c_int), ("rot_mode", C.c_int)]


# name -> (restype, argtypes); mirrors include/vfmreg.h one to one
SIGNATURES = {
    "vfm_last_error": (C.c_char_p, []),
    "vfm_build_info": (C.c_char_p, []),
    "vfm_config_create": (C.c_int, [C.POINTER(c_vp)]),
    "vfm_config_destroy": (C.c_int, [c_vp]),
    "vfm_config_set": (C.c_int, [c_vp, C.c_char_p, c_i64]),
    "vfm_config_get": (C.c_int, [c_vp, C.c_char_p, C.POINTER(c_i64)]),
    "vfm_config_use": (C.c_int, [c_vp]),
    "vfm_l2norm_rows_f32": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_vp]),
    "vfm_match_ip_top1_workspace_bytes": (C.c_size_t, [c_i64, c_i64, C.c_int, C.c_int]),
    "vfm_match_ip_top1": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_match_ip_top1_gated": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, C.c_float, c_vp, c_vp, c_vp, C.c_size_t,
                                          c_vp]),
    "vfm_match_prepared_bytes": (C.c_size_t, [c_i64, C.c_int]),
    "vfm_match_prepare": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_vp]),
    "vfm_match_prepare2": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, C.c_int, c_vp]),
    "vfm_match_prepare2_gated": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, C.c_int, c_vp]),
    "vfm_match_prepare2_gated_p": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, C.c_int, C.c_int, c_vp]),
    "vfm_match_prepare2_gated_t": (C.c_int, [c_vp, C.c_int, c_i64, c_vp, c_vp, C.c_int, c_i64, c_vp, C.c_int, C.c_int, c_vp]),
    "vfm_match_search_finish_gated_t": (C.c_int, [c_vp, C.c_int, c_vp, c_i64, c_vp, C.c_int, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp,
                                                  C.c_size_t, C.c_float, C.c_int, c_vp]),
    "vfm_match_search_workspace_bytes": (C.c_size_t, [c_i64, c_i64, C.c_int]),
    "vfm_match_search_prepared": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp,
                                            C.c_size_t, c_vp]),
    "vfm_match_search_coarse": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, C.c_size_t, c_vp]),
    "vfm_match_search_rescans_async": (C.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "vfm_match_search_probe_half": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, C.c_size_t, C.c_float, c_vp, c_vp]),
    "vfm_match_search_coarse_gated": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, C.c_size_t, c_vp]),
    "vfm_match_search_coarse_gated_r": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, C.c_size_t, C.c_int, c_vp]),
    "vfm_match_search_coarse_gated_g": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, C.c_size_t, C.c_int, C.c_float, c_vp]),
    "vfm_match_search_finish_gated_r": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp,
                                                  C.c_size_t, C.c_float, C.c_int, c_vp]),
    "vfm_match_search_finish": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp,
                                          C.c_size_t, c_vp]),
    "vfm_match_search_finish_gated": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp,
                                                C.c_size_t, C.c_float, c_vp]),
    "vfm_threshold_compact": (C.c_int, [c_vp, c_vp, c_i64, C.c_double, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vfm_match_mutual_l2_workspace_bytes": (C.c_size_t, [c_i64, c_i64, C.c_int, C.c_int, C.c_int]),
    "vfm_match_mutual_l2": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, C.c_int, c_vp, c_vp, c_vp, c_vp, C.c_size_t,
                                      c_vp]),
    "vfm_match_mutual_pairs_workspace_bytes": (C.c_size_t, [c_i64, c_i64, C.c_int]),
    "vfm_match_mutual_pairs": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_ransac_workspace_bytes": (C.c_size_t, [c_i64, C.c_int32]),
    "vfm_ransac_corr": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, C.c_double, C.c_int32, C.c_uint64, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_ransac_corr_bounded": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, C.c_double, C.c_int32, C.c_uint64, c_vp, c_vp, c_vp,
                                          c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_kabsch_batched": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_i64, C.c_double, c_vp, c_vp, c_vp]),
    "vfm_project_workspace_bytes": (C.c_size_t, [c_i64]),
    "vfm_project_pinhole_f64": (C.c_int, [C.c_int, c_vp, c_i64, c_vp, c_vp, C.c_double, c_vp, c_vp, c_i64, c_i64,
                                          c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_lift_multicam": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, C.c_int, c_vp, c_vp, c_vp]),
    "vfm_gather_bilinear_patchgrid": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_vp, c_vp,
                                                c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vfm_transform_xyz_f64": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vfm_voxel_first_workspace_bytes": (C.c_size_t, [c_i64]),
    "vfm_voxel_first": (C.c_int, [c_vp, c_i64, c_i64, C.c_double, C.c_int32, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_voxel_robin_workspace_bytes": (C.c_size_t, [c_i64]),
    "vfm_voxel_robin": (C.c_int, [c_vp, c_i64, c_i64, C.c_double, C.c_int32, C.c_uint32, c_i64, c_vp, c_vp, c_vp, c_vp,
                                  C.c_size_t, c_vp]),
    "vfm_voxel_robin_level": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, C.c_double, C.c_uint32, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "vfm_icp_nearest": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, C.c_int32, C.c_double, C.c_double, c_vp, c_vp, c_vp]),
    "vfm_icp_step_nearest": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int32, C.c_double, C.c_double, c_vp, c_vp, c_vp]),
    "vfm_icp_build_system": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_double, c_vp, c_vp]),
    "vfm_icp_desc_stats": (C.c_int, [c_vp, c_i64, C.c_int32, c_vp, c_vp, c_vp]),
    "vfm_icp_step_nearest_desc": (C.c_int, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, C.c_int32,
                                            C.c_double, C.c_double, c_vp, c_vp, c_vp]),
    "vfm_vit_weights_bytes": (C.c_size_t, [C.POINTER(VitConfig)]),
    "vfm_vit_weights_layout": (C.c_int, [C.POINTER(VitConfig), C.POINTER(c_i64), C.POINTER(c_i64), C.c_int]),
    "vfm_vit_workspace_bytes": (C.c_size_t, [C.POINTER(VitConfig), C.c_int]),
    "vfm_vit_forward": (C.c_int, [C.POINTER(VitConfig), c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp,
                                  C.c_size_t, c_vp]),
}


# include/vfmreg_debug.h: measurement hooks (not part of the drop-in contract)
DEBUG_SIGNATURES = {
    "vfm_prof_events_create": (C.c_int, [C.POINTER(c_vp), C.POINTER(c_vp)]),
    "vfm_prof_arm": (C.c_int, [c_vp, c_vp]),
    "vfm_prof_elapsed_ms": (C.c_int, [c_vp, c_vp, C.POINTER(C.c_float)]),
    "vfm_prof_events_destroy": (C.c_int, [c_vp, c_vp]),
    "vfm_debug_match_stats": (C.c_int, [c_vp, c_i64, c_i64, c_vp]),
    "vfm_debug_i8_rows": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp]),
    "vfm_debug_mx6_rows": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp]),
    "vfm_debug_mx6_half_err": (C.c_int, [c_vp, c_i64, C.c_int, c_vp, c_vp]),
    "vfm_debug_voxel_trace": (C.c_int, [c_vp, c_i64, c_vp]),
    "vfm_debug_ransac_counts": (C.c_int, [c_vp, c_i64, C.c_int, c_vp]),
}


def load() -> C.CDLL:
    """Load the HIP library (after torch, so that both share one libamdhip64 runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python vfm-registration_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    import torch  # noqa: F401  -- loads torch's libamdhip64.so first (same soname => one runtime)
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in list(SIGNATURES.items()) + list(DEBUG_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError => the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _install_tool_names(lib)
    _lib = lib
    return lib


# ---------------------------------------------------------------------------------------------------------------------------------
# Kernel policy (include/vfmreg.h, vfm_config_*): a caller-owned object bound per THREAD; the library keeps no process-global switches.
# ---------------------------------------------------------------------------------------------------------------------------------
import threading as _threading

_tls = _threading.local()


class Config:
    """A vfm_config_t: factory settings + what ``set`` changed.  ``use()`` binds it to the calling thread (every library call made from
    this thread afterwards reads its policy from it); ``with cfg.bound():`` binds it for a block and restores the previous binding."""

    def __init__(self, **settings):
        h = c_vp()
        check(load().vfm_config_create(C.byref(h)), "config_create")
        self._h = h
        for k, v in settings.items():
            self.set(k, v)

    def set(self, key: str, value: int) -> "Config":
        check(load().vfm_config_set(self._h, key.encode(), int(value)), f"config_set({key})")
        return self

    def set_vit_gemm(self, narrow: int, wide: int) -> "Config":
        return self.set("vit_gemm", (int(narrow) << 32) | (int(wide) & 0xFFFFFFFF))

    def get(self, key: str) -> int:
        v = c_i64()
        check(load().vfm_config_get(self._h, key.encode(), C.byref(v)), f"config_get({key})")
        return int(v.value)

    def use(self) -> "Config":
        check(load().vfm_config_use(self._h), "config_use")
        _tls.bound = self
        return self

    def bound(self):
        return using(self)

    def __del__(self):
        try:
            if getattr(_tls, "bound", None) is self:
                _lib.vfm_config_use(None)
                _tls.bound = None
            _lib.vfm_config_destroy(self._h)
        except Exception:
            pass


def current() -> "Config | None":
    """The Config bound to the calling thread (None: factory settings)."""
    return getattr(_tls, "bound", None)


class using:
    """``with using(cfg):`` -- cfg (or None = factory settings) bound to this thread inside the block, the previous binding after it.
    Helper threads that call the library on behalf of a caller bind the caller's ``current()`` this way (vfmreg/vit.py)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def __enter__(self):
        self.prev = current()
        load().vfm_config_use(self.cfg._h if self.cfg is not None else None)
        _tls.bound = self.cfg
        return self.cfg

    def __exit__(self, *exc):
        load().vfm_config_use(self.prev._h if self.prev is not None else None)
        _tls.bound = self.prev
        return False


def thread_config() -> Config:
    """The calling thread's Config, created and bound on first use: what the tools' ``lib.vfm_debug_set_*`` names write to."""
    cfg = current()
    if cfg is None:
        cfg = Config().use()
    return cfg


def _install_tool_names(lib) -> None:
    """Rounds 1 - 5 exported process-global setters (vfm_debug_set_*); ~140 scripts under tools/ and the A/B tests call them as
    ``lib.vfm_debug_set_coarse_variant(43)``.  The library no longer has them: these PYTHON functions of the same names set the calling
    thread's Config (``thread_config()``) -- same effect for a single-threaded script, no effect on any other thread."""
    def setter(key):
        def f(value):
            thread_config().set(key, value)
            return 0
        return f
    lib.vfm_debug_set_coarse_variant = setter("coarse_variant")
    lib.vfm_debug_set_coarse_slices = setter("coarse_slices")
    lib.vfm_debug_set_match_stats = setter("match_stats")
    lib.vfm_debug_set_i8_min_queries = setter("i8_min_queries")
    lib.vfm_debug_set_prep_grid = setter("prep_grid")
    lib.vfm_debug_set_voxel_small = setter("voxel_small")
    lib.vfm_debug_set_ransac_exact_only = setter("ransac_exact_only")

    def vit_gemm(narrow, wide):
        thread_config().set_vit_gemm(narrow, wide)
        return 0
    lib.vfm_debug_set_vit_gemm = vit_gemm


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vfm_last_error().decode()
        raise RuntimeError(f"libvfmreg_hip {what} failed ({rc}): {msg}")
