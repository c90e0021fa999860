"""Mirror of the hot-path helper of vfm_reg.utils: ``transform_pcl`` (vfm_reg/utils.py:47-54)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def transform_pcl(pcl: np.ndarray, transform: np.ndarray) -> np.ndarray:
    """xyz' = T @ [xyz; 1] in fp64 on the GPU, descriptors carried through, cast back to pcl.dtype."""
    assert transform.shape == (4, 4), "Invalid shape"
    xyz = torch.from_numpy(np.ascontiguousarray(pcl[:, :3], dtype=np.float64)).cuda()
    T = torch.from_numpy(np.ascontiguousarray(transform, dtype=np.float64)).cuda()
    out = ops.transform_xyz(xyz, T).cpu().numpy()
    pcl_out = np.c_[out, pcl[:, 3:]]
    assert pcl_out.shape == pcl.shape
    return pcl_out.astype(pcl.dtype)
