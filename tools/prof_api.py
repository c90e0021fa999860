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
p = synth.make_pair(int(sys.argv[1]) if len(sys.argv) > 1 else 6000, int(sys.argv[2]) if len(sys.argv) > 2 else 30000, 384, seed=11)
ICP = (sys.argv[3] == "1") if len(sys.argv) > 3 else True
voxel_map = np.c_[p["b_xyz"], p["b_desc"]].astype(np.float32)
raw_scan = np.c_[p["q_xyz"], p["q_desc"]].astype(np.float32)
node = RegistrationNode(cache_map=True)
node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=ICP)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): node.ransac_registration(voxel_map, raw_scan, "vfm", run_icp=ICP)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
