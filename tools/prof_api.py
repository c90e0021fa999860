import sys, cProfile, pstats
from pathlib import Path
import numpy as np
ROOT = Path("/root/repo")
sys.path.insert(0, str(ROOT / "vfm-registration_amd")); sys.path.insert(0, str(ROOT))
import torch
from vfmreg import synth
from vfmreg.mapping import VoxelHashMap
from vfmreg.registration import RegistrationNode
VoxelHashMap.quiet = True
p = synth.make_pair(6000, 30000, 384, seed=11)
voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
node = RegistrationNode()
node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
