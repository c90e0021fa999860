"""The constants of kiss_icp.config the registration path reads (config.py:28-43, parser.py:61-84):
``load_config(None, None)`` of the reference yields voxel_size = max_range / 100 = 1.0,
max_points_per_voxel = 20, max_range = 100, initial_threshold = 2.0."""
from types import SimpleNamespace

DESCRIPTOR_SIZE = 384             # descriptor_size.hpp:7
POINT_SIZE = 3 + DESCRIPTOR_SIZE  # descriptor_size.hpp:13, kiss_icp_pybind._point_size()


def load_config(config_file=None, deskew=None, max_range=None):
    max_range = 100.0 if max_range is None else float(max_range)
    return SimpleNamespace(
        data=SimpleNamespace(max_range=max_range, min_range=5.0, deskew=bool(deskew) if deskew is not None else False,
                             preprocess=True),
        mapping=SimpleNamespace(voxel_size=float(max_range / 100.0), max_points_per_voxel=20),
        adaptive_threshold=SimpleNamespace(fixed_threshold=None, initial_threshold=2.0, min_motion_th=0.1),
    )
