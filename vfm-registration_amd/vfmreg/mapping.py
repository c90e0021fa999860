"""Mirror of kiss_icp.mapping (src/kiss-icp/python/kiss_icp/mapping.py:30-131) for the calls the
registration path makes: ``get_voxel_hash_map``, ``VoxelHashMap.add_points``, ``.point_cloud``,
``.point_cloud_n`` and ``.get_vfm_correspondences`` (C++: VoxelHashMap.cpp:461-626, 628-676,
733-770; binding kiss_icp_pybind.cpp:75-129).

The map keeps at most ``max_points_per_voxel`` points per voxel in insertion order
(VoxelHashMap.hpp:55-62).  ``point_cloud*()`` and the row numbering of the descriptor search follow
the reference's container: voxels in ``tsl::robin_map`` iteration order (a default-constructed map
that grew by doubling, VoxelHash of VoxelHashMap.hpp:72-77), the points of a voxel in insertion
order (csrc/voxel.hip ``vfm_voxel_robin``).  The map state is a pure function of the sequence of
kept points, so it is recomputed from that sequence when points were added.  The descriptor search
runs on the GPU through the C ABI; besides the reference's (source, target) coordinate pair the
indices are available (``get_vfm_correspondence_indices``), which makes the KD-tree index recovery
of registration_node.py:288-317 unnecessary.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops
from .voxelization import first_per_voxel, robin_order


def get_voxel_hash_map(config):
    return VoxelHashMap(voxel_size=config.mapping.voxel_size, max_distance=config.data.max_range,
                        max_points_per_voxel=config.mapping.max_points_per_voxel)


class VoxelHashMap:
    quiet = False  # the reference prints a stats line per search (VoxelHashMap.cpp:613-616)

    def __init__(self, voxel_size: float, max_distance: float, max_points_per_voxel: int):
        self.voxel_size = float(voxel_size)
        self.max_distance = float(max_distance)
        self.max_points_per_voxel = int(max_points_per_voxel)
        self.clear()

    # ------------------------------------------------------------------ container
    def clear(self):
        self._chunks = {}       # kind -> list of arrays, in insertion order (3-D and N-D points live in separate maps)
        self._ordered = {}      # kind -> rows in container iteration order (cache)
        self._dev = None        # cached device copy of the N-D map (IndexFlatIP.add)

    @staticmethod
    def _kind(width: int) -> str:
        # mapping.py:77-85 routes 3 columns to map_, _point_size() (= 3 + DESCRIPTOR_SIZE, a compile-time 384 in
        # DescriptorSize.hpp:7) to map_n_ and other widths to map_x_.  Here the descriptor width is a run-time
        # property (configs C3 and C5 use 384 and 768): the first wide insert fixes it, a different width later is
        # an error, and map_x_ (never read by the path) does not exist.
        return "3" if width == 3 else "n"

    def empty(self):
        return not self._chunks.get("3")

    def empty_n(self):
        return not self._chunks.get("n")

    def add_points(self, points: np.ndarray):
        points = np.asarray(points)
        if points.ndim != 2 or points.shape[1] < 3:
            raise ValueError("Invalid shape")  # mapping.py:86
        kind = self._kind(points.shape[1])
        pts = np.ascontiguousarray(points, dtype=np.float64)  # pybind: forcecast to double (stl_vector_eigen.h:73-86)
        if len(pts) == 0:
            return
        # VoxelBlock::AddPoint keeps a point iff its voxel holds fewer than max_points_per_voxel points
        # (VoxelHashMap.hpp:55-62).  The points already stored come first and are all within the cap,
        # so "first K per voxel of [stored..., new...]" restricted to the new rows is exactly that rule.
        stored = self._chunks.get(kind, [])
        if stored and stored[0].shape[1] != pts.shape[1]:
            raise ValueError("Invalid shape")
        n_old = sum(len(a) for a in stored)
        xyz = np.concatenate([a[:, :3] for a in stored] + [pts[:, :3]], axis=0) if n_old else pts[:, :3]
        keep = first_per_voxel(xyz, self.voxel_size, self.max_points_per_voxel)
        keep_new = keep[keep >= n_old] - n_old
        self._chunks.setdefault(kind, []).append(pts[keep_new])
        self._ordered.pop(kind, None)
        self._dev = None

    def _cloud(self, kind: str) -> Optional[np.ndarray]:
        """Rows of one of the three maps as the reference walks it (VoxelHashMap.cpp:628-676)."""
        arrs = self._chunks.get(kind)
        if not arrs:
            return None
        if kind not in self._ordered:
            rows = np.concatenate(arrs, axis=0)
            # every stored row is a kept one: replaying them reproduces the container (voxels are created
            # by their first point, which is always kept)
            order = robin_order(rows, self.voxel_size, self.max_points_per_voxel, reserve=False, hash_mul=ops.HASH_MAP)
            assert len(order) == len(rows)
            self._ordered[kind] = rows[order]
        return self._ordered[kind]

    def point_cloud(self) -> np.ndarray:
        for kind in ("3", "n"):  # VoxelHashMap.cpp:631-659: map_, else map_n_ (else map_x_)
            c = self._cloud(kind)
            if c is not None:
                return np.ascontiguousarray(c[:, :3])
        return np.zeros((0, 3))

    def point_cloud_n(self) -> np.ndarray:
        c = self._cloud("n")  # VoxelHashMap.cpp:664-676
        return c if c is not None else np.zeros((0, 3))

    # ------------------------------------------------------------------ search
    def _device_map(self):
        if self._dev is None:
            m = self.point_cloud_n()
            desc = torch.from_numpy(np.ascontiguousarray(m[:, 3:], dtype=np.float32)).cuda()  # VoxelHashMap.cpp:472-473
            xyz = torch.from_numpy(np.ascontiguousarray(m[:, :3])).cuda()
            self._dev = (desc, xyz)
        return self._dev

    def get_vfm_correspondence_indices(self, points: np.ndarray, min_cosine_similarity: float):
        """(query_idx[K], map_idx[K], sim[N]) -- the indices behind get_vfm_correspondences."""
        points = np.asarray(points)
        b_desc, _ = self._device_map()
        if points.ndim != 2 or points.shape[1] != b_desc.shape[1] + 3:
            raise RuntimeError("Unable to cast Python instance to C++ type: expected %d columns"
                               % (b_desc.shape[1] + 3))  # py::cast_error, stl_vector_eigen.h:76-78
        q_desc = torch.from_numpy(np.ascontiguousarray(points[:, 3:], dtype=np.float32)).cuda()
        d = q_desc.shape[1]
        prec = ops.FAST if (d % 128 == 0 and 128 <= d <= 768) else ops.EXACT
        idx, sim = ops.match_ip_top1(q_desc, b_desc, prec)
        r = ops.threshold_compact(sim, idx, float(min_cosine_similarity), want_corres=True)
        k = int(r["count"].item())
        corres = r["corres"][:k].cpu().numpy()
        return corres[:, 0].astype(np.int64), corres[:, 1].astype(np.int64), sim.cpu().numpy()

    def get_vfm_correspondences(self, points: np.ndarray, max_correspondance_distance: float
                                ) -> Tuple[np.ndarray, np.ndarray]:
        """Pair of {source, target} coordinates (mapping.py:120-131); the float is the cosine
        threshold despite its name (kiss_icp_pybind.cpp:128-129)."""
        points = np.asarray(points)
        qi, mi, sim = self.get_vfm_correspondence_indices(points, max_correspondance_distance)
        m = self.point_cloud_n()
        if not self.quiet:
            print(f"Points: {len(points)} | Corrs.: {len(qi)} | Outliers: 0 | Mean sim.: {float(sim.mean()) if len(sim) else 0.0}")
        return np.asarray(points[qi, :3], dtype=np.float64), np.asarray(m[mi, :3], dtype=np.float64)
